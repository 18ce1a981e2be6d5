#!/usr/bin/env python3
"""bench.py — the hot path's headline metric (BASELINE.json): aggregate env-steps/s.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: under python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ..., or on its own:
     without WORLD_SIZE in the environment it starts the N ranks itself, one child process per GPU, before touching a GPU)

A "step" is one SMARTS tick of every environment instance of the shard: controllers, vehicle
dynamics, collisions, sensors/observation build, events/reward/done and auto-reset — one
``smx_step`` call, which enqueues the tick's kernels (k_control, k_scan, [k_ogm], k_sensors, k_commit
and, with auto-reset, the reset pass) on one stream.
Workload: BASELINE configs[3], the configuration the metric is quoted on — scenarios/loop, 4096 batched
envs x 32 Laner agents, waypoints (4, 20) + neighbourhood (10, 50 m) + OGM 64 x 64 observations, dt = 0.1 s,
synthetic spawns / action stream per SURVEY.md §8d.  ``--gpus N`` shards those 4096 envs, 4096 / N per rank
(strong scaling, no data-path collective, only the small reward/done gather); ``--scaling weak`` gives every
rank all 4096.  ``--config c2|c3|c5`` selects the other BASELINE configurations for profiling.
Inputs (state, spawn table, action stream) are resident in HBM when the timed region starts.

After the timed region a short second pass (not timed, not part of ``value``) re-runs the tick
with a boundary event after every kernel to attribute the time per kernel (``roofline.kernels``)
and to report obs-build ms/tick.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable)
ACTION_CYCLE = 64
TIMED_EVERY = 4  # launches between two event-timed ones inside the timed region


def action_stream(E, N, seed, first_env):
    """[ACTION_CYCLE, E, N] int8: keep_lane w.p. 0.8 else uniform over the other three (§8d),
    drawn per env from PCG64(seed + global env index) so a shard reproduces its slice."""
    out = np.zeros((ACTION_CYCLE, E, N), dtype=np.int8)
    for e in range(E):
        rng = np.random.Generator(np.random.PCG64(seed + first_env + e + 1_000_003))
        u = rng.random((ACTION_CYCLE, N))
        other = rng.integers(1, 4, (ACTION_CYCLE, N))
        out[:, e] = np.where(u < 0.8, 0, other)
    return out


# BASELINE.json configs[1..4]: scenario, envs per GPU, vehicles per env, extra sensors
CONFIGS = {
    "c2": dict(scenario="loop", envs=1024, vehicles=8, extra={}, label="configs[1]"),
    "c3": dict(scenario="intersections/4lane", envs=2048, vehicles=16, extra={}, label="configs[2]"),
    "c4": dict(scenario="loop", envs=4096, vehicles=32,
               extra=dict(ogm=True, ogm_width=64, ogm_height=64, ogm_resolution=50 / 64), label="configs[3]"),
    "c5": dict(scenario="minicity", envs=4096, vehicles=64, extra=dict(lidar="planar100"), label="configs[4]"),
}


def hbm_traffic_for(workload_key):
    """HBM bytes per smx_step from the committed PMC passes (profiles/*_hbm_traffic.json, written
    by tools/collect_hbm_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of
    this same command).  None when no pass was recorded for this workload."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")), reverse=True):
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        if rec.get("workload_key") == workload_key:
            return rec
    return None


def cpu_port_rate(net, cm, cfg_kw, seconds_budget, first_env=0):
    """(env-steps, seconds, envs, agents, ticks) of the CPU port (oracle/: per-agent sequential
    Python/numpy, the shape of SMARTS._step) on a bounded sample of the workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity
    from smarts_amd.engine import SimConfig, make_spawns

    N = cfg_kw["num_vehicles"]
    E = 4 if N <= 8 else 1
    kw = dict(cfg_kw)
    kw["num_envs"] = E
    cfg = SimConfig(**kw)
    spawns = make_spawns(cm, E, N, episodes=1, seed=42, first_env=first_env)
    ob = parity.OracleBatch(net, cm, cfg, spawns[0])
    ob.reset_observe()
    acts = action_stream(E, N, 42, first_env)
    t0 = time.perf_counter()
    ticks = 0
    while True:
        ob.step(acts[ticks % ACTION_CYCLE])
        ticks += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or ticks >= 400:
            break
    return E * ticks, el, E, N, ticks


def workload_config(config, envs=None, vehicles=None, scenario=None):
    from smarts_amd import lidar as lidar_mod

    preset = CONFIGS[config]
    E, N = envs or preset["envs"], vehicles or preset["vehicles"]
    cfg_kw = dict(num_envs=E, num_vehicles=N, dt=0.1, waypoints=True, neighbors=True, nb_radius=50.0, nb_max=10,
                  wp_paths=4, wp_len=20, wp_lookahead=32, auto_reset=True)
    cfg_kw.update(preset["extra"])
    if cfg_kw.get("lidar") == "planar100":
        cfg_kw["lidar"] = lidar_mod.Planar100
    return preset, scenario or preset["scenario"], cfg_kw


def cpu_worker(args):
    """Child of the all-cores CPU leg: one process = one env group, as ParallelEnv runs them
    (parallel_env.py:96-122).  Never touches the GPU (no torch import)."""
    from smarts_amd.map_compiler import compile_map
    from smarts_amd.sumo_map import load_net

    _, scenario, cfg_kw = workload_config(args.config, args.envs_per_gpu, args.vehicles, args.scenario)
    net = load_net(os.path.join(ROOT, "smarts_amd", "scenarios", scenario))
    steps, el, *_ = cpu_port_rate(net, compile_map(net), cfg_kw, args.cpu_seconds, first_env=4 * args.cpu_worker)
    print(json.dumps({"env_steps": steps, "seconds": el}))


def cpu_all_cores(args, seconds_budget=8.0):
    """P processes x their own envs, P = the host cores this job may use
    (at most 16, the GPU box's share).  Children are started as ordinary subprocesses with the
    GPU hidden and no profiler preload, and only after this process has finished its GPU work."""
    import subprocess

    if any("rocprof" in v.lower() for v in (os.environ.get("LD_PRELOAD", ""), os.environ.get("ROCP_TOOL_LIBRARIES", ""))):
        return None  # under the profiler every child would attach to the GPU
    P = max(1, min(len(os.sched_getaffinity(0)), 16))
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF"))}
    env.update(HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1",
               MKL_NUM_THREADS="1")
    cmd = [sys.executable, os.path.abspath(__file__), "--config", args.config, "--cpu-seconds", str(seconds_budget)]
    for flag, val in (("--envs-per-gpu", args.envs_per_gpu), ("--vehicles", args.vehicles), ("--scenario", args.scenario)):
        if val is not None:
            cmd += [flag, str(val)]
    procs = [subprocess.Popen(cmd + ["--cpu-worker", str(w)], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
             for w in range(P)]
    rate = 0.0
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=seconds_budget * 6 + 60)
            rec = json.loads(out.decode().strip().splitlines()[-1])
            rate += rec["env_steps"] / rec["seconds"]
        except Exception:
            pr.kill()
            return None
    return {"value": rate, "unit": "env-steps/s", "cores": P,
            "sample": f"{P} processes, each the single-core sample on its own envs for ~{seconds_budget:.0f} s, rates summed"}


def cpu_baseline(net, cm, cfg_kw, args, seconds_budget=15.0):
    """The CPU port timed on a bounded sample of the same workload.  Reported, never shipped."""
    steps, el, E, N, ticks = cpu_port_rate(net, cm, cfg_kw, seconds_budget)
    return {
        "value": steps / el,
        "unit": "env-steps/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{E} envs x {N} agents x {ticks} ticks of the same workload (same map, sensors, spawns and "
                  f"action stream), {el:.1f} s on one host core; the reference itself is single-threaded Python "
                  "per env",
        "all_cores": cpu_all_cores(args),
    }


def copy_peak_gbps(device, torch):
    """Measured device-to-device copy rate (read + write bytes / time) and fill rate (write bytes /
    time) of a 1 GiB buffer: the practical HBM ceilings on this box, quoted beside the datasheet
    peak."""
    n = 1 << 30
    src = torch.empty(n, dtype=torch.uint8, device=device)
    dst = torch.empty_like(src)
    for _ in range(3):
        dst.copy_(src)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        dst.copy_(src)
    b.record()
    torch.cuda.synchronize()
    copy = 2.0 * n * 10 / (a.elapsed_time(b) * 1e-3) / 1e9
    # write-only stream (the path is write-dominated: dense observation rows)
    for _ in range(3):
        dst.zero_()
    a.record()
    for _ in range(10):
        dst.zero_()
    b.record()
    torch.cuda.synchronize()
    fill = 1.0 * n * 10 / (a.elapsed_time(b) * 1e-3) / 1e9
    del src, dst
    return copy, fill


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--repeats", type=int, default=3, help="timed regions of --steps ticks each; the median is reported")
    ap.add_argument("--config", default="c4", choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default="strong", choices=("strong", "weak"),
                    help="strong: the configuration's envs are sharded over the ranks (4096/N each); weak: every rank runs them all")
    ap.add_argument("--envs-per-gpu", type=int, default=None, help="override: this many envs on every rank (weak scaling)")
    ap.add_argument("--vehicles", type=int, default=None)
    ap.add_argument("--scenario", default=None)
    ap.add_argument("--phase-steps", type=int, default=100)
    ap.add_argument("--launch-strategy", default="auto", choices=("auto", "small", "large", "large_one_lane", "large_teams"),
                    help="how a tick is cut into launches (include/smx.h); auto = by vehicle count")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-worker", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help=argparse.SUPPRESS)
    ap.add_argument("--launch-check", action="store_true",
                    help="only rendezvous (gloo, no GPU), all-reduce the ranks and print {n_gpus, rank_sum}: the CPU test of the launch path")
    args = ap.parse_args()
    if args.cpu_worker is not None:
        return cpu_worker(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks here, before anything in this process
        # has touched the GPU (torch is not even imported yet); rank 0 prints the line
        from smarts_amd.sharding import self_launch

        raise SystemExit(self_launch([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    if args.launch_check:
        os.environ.setdefault("SMX_DIST_BACKEND", "gloo")
        import torch
        import torch.distributed as dist

        from smarts_amd import sharding

        rank, _, world = sharding.init_process_group()
        t = torch.tensor([float(rank)])
        if world > 1:
            dist.all_reduce(t)
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "rank_sum": float(t.item())}))
        return

    import torch

    from smarts_amd import build, sharding
    from smarts_amd.engine import BatchedSim, SimConfig, make_spawns
    from smarts_amd.map_compiler import compile_map
    from smarts_amd.sumo_map import load_net
    import torch.distributed as dist

    rank, local_rank, world = sharding.init_process_group()
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the product path has no CPU fallback)")
    dev_index = sharding.local_device_index(local_rank)
    torch.cuda.set_device(dev_index)
    device = f"cuda:{dev_index}"
    if rank == 0:
        build.build()
    if world > 1:
        sharding.barrier()

    preset, scenario, cfg_kw = workload_config(args.config, args.envs_per_gpu, args.vehicles, args.scenario)
    N = cfg_kw["num_vehicles"]
    # The job: the configuration's env instances, sharded one contiguous range per GPU (the
    # reference's own data parallelism is one process per env, parallel_env.py:96-122, seeds seed + global index
    # :190-202).  Strong scaling by default: 4096 envs in all, 4096 / N per rank.
    weak = args.scaling == "weak" or args.envs_per_gpu is not None
    total_envs = cfg_kw["num_envs"] * world if weak else cfg_kw["num_envs"]
    plan = sharding.ShardPlan(total_envs=total_envs, world_size=world, rank=rank)
    E = plan.num_envs
    cfg_kw["num_envs"] = E
    net = load_net(os.path.join(ROOT, "smarts_amd", "scenarios", scenario))
    cm = compile_map(net)
    cfg = SimConfig(launch_strategy=args.launch_strategy, **cfg_kw)
    spawns = make_spawns(cm, E, N, episodes=4, seed=42, first_env=plan.first_env)
    sim = BatchedSim(cm, cfg, device=device, spawns=spawns)
    actions = torch.from_numpy(action_stream(E, N, 42, plan.first_env)).to(device)
    equal_shards = total_envs % world == 0  # the gather is an all_gather: equal blocks only
    gather = sharding.RewardDoneGather(E, N, device, world if equal_shards else 1)

    sim.reset()

    def tick(i):
        gather.release(sim.next_learner_block)  # the gather of two ticks ago has read it
        o = sim.step(actions[i % ACTION_CYCLE])
        gather.start_packed(o["learner"])  # learner-side {reward, done} block; overlaps the next tick

    for i in range(args.warmup):
        tick(i)
    gather.finish()
    # Timed regions: EXACTLY --steps ticks each, bracketed by barrier + synchronize on both sides, the MAX over
    # ranks taken per region; --repeats of them back to back, the median one is the reported value (§8d).
    # HIP events around the launch sequence of every TIMED_EVERY-th step (an event pair costs the stream ~6 us).
    region_s = []
    alive_samples = []  # agents with a vehicle (rows being written), sampled outside the timed regions
    done_ticks = args.warmup
    for rep in range(max(1, args.repeats)):
        torch.cuda.synchronize()
        alive_samples.append(float(sim.out["active"].sum().item()))
        if world > 1:
            sharding.barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            sampled = i % TIMED_EVERY == 0
            if sampled:
                sim.set_timing(True)
            tick(done_ticks + i)
            if sampled:
                sim.set_timing(False)
        gather.finish()  # every tick's gather has landed inside the timed region
        torch.cuda.synchronize()
        if world > 1:
            sharding.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        region_s.append(el)
        alive_samples.append(float(sim.out["active"].sum().item()))
        done_ticks += args.steps
    # alive agents of the whole job (every rank's shard)
    alive_job = float(np.mean(alive_samples))
    if world > 1:
        t = torch.tensor([alive_job], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        alive_job = float(t.item())
    elapsed = float(np.median(region_s))
    kernel_ms = sim.read_step_ms()

    # per-kernel attribution pass (outside the timed region)
    phase_ms = None
    if rank == 0 and args.phase_steps > 0:
        sim.set_timing(2)
        for i in range(args.phase_steps):
            sim.step(actions[(done_ticks + i) % ACTION_CYCLE])
        torch.cuda.synchronize()
        phase_ms = sim.read_phase_ms().mean(axis=0)
        sim.set_timing(0)

    copy_peak, fill_peak = copy_peak_gbps(device, torch) if rank == 0 else (None, None)
    if rank == 0:
        env_steps_per_s = total_envs * args.steps / elapsed
        # algorithmic bytes of one launch (this rank's shard): SURVEY.md §8(d) per agent-step x E x N
        kb = sim.kernel_bytes_per_agent_step()
        read_agent, write_agent = sum(r for r, _ in kb.values()), sum(w for _, w in kb.values())
        bytes_agent = read_agent + write_agent
        # Agents whose vehicle is gone write no rows and run no controller until their env restarts (the env
        # restarts when ALL its agents are done, parallel_env.py:303-309): the algorithmic bytes of a tick are
        # those of the agents alive in it.  Alive agents are sampled at both ends of every timed region.
        alive_mean = float(np.mean(alive_samples))
        agents = alive_mean
        avg_kernel_s = float(np.median(kernel_ms)) * 1e-3
        achieved = bytes_agent * agents / avg_kernel_s / 1e9
        workload_key = f"{args.config}:{scenario}:{E}x{N}"
        traffic = hbm_traffic_for(workload_key)
        kernels = None
        obs_build_ms = None
        dominant = None
        if phase_ms is not None:
            from smarts_amd._native import PHASES

            kernels = {}
            for name, ms in zip(PHASES, phase_ms):
                ms = float(ms)
                if name == "ogm" and name not in kb:
                    continue
                r, w = kb.get(name, (0, 0))
                b = (r + w) * agents
                kernels["k_" + name if name != "reset" else "reset_pass"] = {
                    "avg_ms": ms, "algorithmic_bytes": b, "read_bytes": r * agents, "write_bytes": w * agents,
                    "GB/s": (b / (ms * 1e-3) / 1e9) if ms > 0 and b else None,
                }
            obs_build_ms = float(sum(ms for n_, ms in zip(PHASES, phase_ms) if n_ in ("scan", "ogm", "sensors", "commit")))
            # the dominant kernel of an HBM-bound path is the one that moves the most algorithmic bytes
            dominant = max((k for k in kernels if k != "reset_pass"), key=lambda k: kernels[k]["algorithmic_bytes"])
        sensors = ("waypoints(4x20, lookahead 32)+neighbours(10, r=50 m) obs"
                   + (", OGM 64x64 @ 50/64 m/px" if cfg.ogm else "")
                   + (", lidar 100 rays x 20 m" if cfg.lidar is not None else ""))
        line = {
            "metric": "aggregate env-steps/s (all agents)",
            "value": env_steps_per_s,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if weak else "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "repeats": {"n": len(region_s), "reported": "median",
                        "env_steps_per_s": [total_envs * args.steps / t for t in region_s]},
            "config": {
                "workload": f"scenarios/{scenario}, {total_envs} batched envs x {N} Laner agents, {sensors}, dt=0.1, "
                            f"auto-reset (BASELINE.json {preset['label']})",
                "total_envs": total_envs,
                "envs_per_gpu": E,
                "vehicles_per_env": N,
                # agent-steps of agents that HAVE a vehicle (rows written, controller run): slots whose agent is done
                # until its env restarts are not counted (value x N would count them)
                "alive_agent_steps_per_s": alive_job * args.steps / elapsed,
                "alive_fraction": alive_job / (total_envs * N),
                "alive_agents_per_tick": {"mean_rank0": alive_mean, "of": E * N, "mean_job": alive_job,
                                          "note": "an env restarts when all its agents are done; until then the "
                                                  "agents already done have no vehicle: rows and byte counts "
                                                  "are those of the alive agents"},
                "launch_form": sim.launch_form(),
                "sharding": f"{world} rank(s), envs [g*{total_envs}/{world}, (g+1)*{total_envs}/{world}) on GPU g "
                            f"({E} on rank 0); no data-path collective; per-tick reward/done all_gather (RCCL)",
                "obs_build_ms_per_tick": obs_build_ms,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "achieved_read": read_agent * agents / avg_kernel_s / 1e9,
                "achieved_write": write_agent * agents / avg_kernel_s / 1e9,
                "algorithmic_bytes": {"read": read_agent * agents, "write": write_agent * agents,
                                      "total": bytes_agent * agents, "per_agent_step": bytes_agent,
                                      "handoff_not_counted": sim.handoff_bytes_per_agent_step() * agents},
                "traffic": traffic["bytes_per_step"] if traffic else None,
                "traffic_read": traffic["read_bytes_per_step"] if traffic else None,
                "traffic_write": traffic["write_bytes_per_step"] if traffic else None,
                "traffic_source": traffic["source"] if traffic else None,
                "measured_copy_peak": copy_peak,
                "measured_fill_peak": fill_peak,
                "kernel": "smx_step of rank 0's shard: the tick's launch sequence k_control > k_scan > [k_ogm] > "
                          "k_waypoints + k_observe [+ k_lidar] (one k_sensors launch on small batches) > k_commit > "
                          "reset pass; duration = HIP events around the sequence on its stream, median over the "
                          "sampled launches of the timed regions",
                "avg_kernel_ms": avg_kernel_s * 1e3,
                "timed_launches": int(len(kernel_ms)),
                "dominant_kernel": dominant,
                "kernels": kernels,
            },
        }
        if not args.no_cpu_baseline and world == 1:
            sim.close()
            sim = None
            torch.cuda.synchronize()
            line["cpu_baseline"] = cpu_baseline(net, cm, cfg_kw, args)
        elif world == 1:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if sim is not None:
        sim.close()
    if world > 1:
        sharding.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
