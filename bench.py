#!/usr/bin/env python3
"""bench.py — the hot path's headline metric (BASELINE.json): aggregate env-steps/s.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one SMARTS tick of every environment instance of the shard: controllers, vehicle
dynamics, collisions, sensors/observation build, events/reward/done and auto-reset — one
``smx_step`` launch.  Workload at every N: BASELINE config[1] per GPU — scenarios/loop, 1024
batched envs x 8 Laner agents, waypoints (4, 20) + neighbourhood (10, 50 m) observations,
dt = 0.1 s, synthetic spawns / action stream per SURVEY.md §8d (weak scaling: each rank owns
its own 1024 envs; no data-path collective, only the small reward/done gather).
Inputs (state, spawn table, action stream) are resident in HBM when the timed region starts.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable)
ACTION_CYCLE = 64


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


def cpu_baseline(net, cm, cfg_kw, seconds_budget=15.0):
    """The CPU port (oracle/: per-agent sequential Python/numpy, the shape of SMARTS._step) timed
    on a bounded sample of the same workload.  Reported, never shipped."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity
    from smarts_amd.engine import SimConfig, make_spawns

    E, N = 4, cfg_kw["num_vehicles"]
    kw = dict(cfg_kw)
    kw["num_envs"] = E
    cfg = SimConfig(**kw)
    spawns = make_spawns(cm, E, N, episodes=1, seed=42)
    ob = parity.OracleBatch(net, cm, cfg, spawns[0])
    ob.reset_observe()
    acts = action_stream(E, N, 42, 0)
    t0 = time.perf_counter()
    ticks = 0
    while True:
        ob.step(acts[ticks % ACTION_CYCLE])
        ticks += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or ticks >= 400:
            break
    return {
        "value": E * ticks / el,
        "unit": "env-steps/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{E} envs x {N} agents x {ticks} ticks of the same workload (scenarios/loop, waypoints+neighbours), "
                  f"{el:.1f} s on one host core; the reference itself is single-threaded Python per env",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs-per-gpu", type=int, default=1024)
    ap.add_argument("--vehicles", type=int, default=8)
    ap.add_argument("--scenario", default="loop")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from smarts_amd import build, sharding
    from smarts_amd.engine import BatchedSim, SimConfig, make_spawns
    from smarts_amd.map_compiler import compile_map
    from smarts_amd.sumo_map import load_net
    import torch.distributed as dist

    rank, local_rank, world = sharding.init_process_group()
    assert world == args.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if rank == 0:
        build.build()
    if world > 1:
        dist.barrier()

    E, N = args.envs_per_gpu, args.vehicles
    plan = sharding.ShardPlan(total_envs=E * world, world_size=world, rank=rank)
    net = load_net(os.path.join(ROOT, "smarts_amd", "scenarios", args.scenario))
    cm = compile_map(net)
    cfg_kw = dict(num_envs=E, num_vehicles=N, dt=0.1, waypoints=True, neighbors=True, nb_radius=50.0, nb_max=10,
                  wp_paths=4, wp_len=20, wp_lookahead=32, auto_reset=True)
    cfg = SimConfig(**cfg_kw)
    spawns = make_spawns(cm, E, N, episodes=4, seed=42, first_env=plan.first_env)
    sim = BatchedSim(cm, cfg, device=device, spawns=spawns)
    actions = torch.from_numpy(action_stream(E, N, 42, plan.first_env)).to(device)
    gather = sharding.RewardDoneGather(E, N, device, world)

    out = sim.reset()

    def tick(i):
        o = sim.step(actions[i % ACTION_CYCLE])
        gather(o["reward"], o["done"])

    for i in range(args.warmup):
        tick(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    sim.set_timing(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        tick(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    sim.set_timing(False)
    kernel_ms = sim.read_step_ms()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_envs = E * world
        env_steps_per_s = total_envs * args.steps / elapsed
        # algorithmic bytes of one launch: per agent-step, state read + state write + action +
        # every observation / reward / done byte written (dense StdObs layout)
        bytes_agent = 2 * sim.state_bytes_per_agent_step() + 1 + sim.output_bytes_per_agent_step()
        bytes_launch = bytes_agent * E * N
        avg_kernel_s = float(np.mean(kernel_ms)) * 1e-3
        achieved = bytes_launch / avg_kernel_s / 1e9
        line = {
            "metric": "aggregate env-steps/s (all agents)",
            "value": env_steps_per_s,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"scenarios/{args.scenario}, {E} batched envs x {N} Laner agents per GPU, "
                            "waypoints(4x20, lookahead 32)+neighbours(10, r=50 m) obs, dt=0.1, auto-reset "
                            "(BASELINE.json configs[1])",
                "envs_per_gpu": E,
                "vehicles_per_env": N,
                "agent_steps_per_s": env_steps_per_s * N,
                "sharding": f"{world} x ({E} envs), no data-path collective; per-tick reward/done all_gather",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": None,
                "kernel": "smx_tick_kernel",
                "avg_kernel_ms": avg_kernel_s * 1e3,
                "bytes_per_agent_step": bytes_agent,
            },
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(net, cm, cfg_kw)
        elif world == 1:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    sim.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
