"""Developer: time library variants (built beforehand with smarts_amd.build.build_variant, -D switches) on one box.
    python tools/dev_variants.py c4 <variant.so> [<variant.so> ...]
For every variant, in a child process with $SMX_LIBRARY set: the per-kernel phase times of a serial tick
(smx_set_timing(2)) over ticks 10-70, and the forked tick's wall time over ticks 5-65 (the driver's window)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, time, json
sys.path.insert(0, %(root)r)
import torch, bench
from smarts_amd.engine import BatchedSim, SimConfig, make_spawns
from smarts_amd.map_compiler import compile_map
from smarts_amd.sumo_map import load_net
config = %(config)r
preset, scenario, cfg_kw = bench.workload_config(config)
if %(envs)r: cfg_kw["num_envs"] = %(envs)r
E, N = cfg_kw["num_envs"], cfg_kw["num_vehicles"]
cm = compile_map(load_net(os.path.join(%(root)r, "smarts_amd", "scenarios", scenario)))
spawns = make_spawns(cm, E, N, episodes=4, seed=42)
actions = torch.from_numpy(bench.action_stream(E, N, 42, 0)).cuda()
def run(timing, warm, steps):
    sim = BatchedSim(cm, SimConfig(**cfg_kw), spawns=spawns)
    sim.reset()
    for i in range(warm):
        sim.step(actions[i %% bench.ACTION_CYCLE])
    torch.cuda.synchronize()
    if timing: sim.set_timing(2)
    t0 = time.perf_counter()
    for i in range(steps):
        sim.step(actions[(warm + i) %% bench.ACTION_CYCLE])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    alive = float(sim.out["active"].sum().item()) / (E * N)
    ph = sim.read_phase_ms().mean(axis=0).round(4).tolist() if timing else None
    chk = float(sim.state[:3].sum().item())
    sim.close()
    return el / steps * 1e3, ph, alive, chk
ms_f, _, alive, chk = run(False, 5, 60)
ms_f2, _, _, _ = run(False, 5, 60)
_, ph, _, _ = run(True, 10, 60)
print(json.dumps(dict(tick_ms=min(ms_f, ms_f2), phases=dict(zip(["control", "scan", "ogm", "sensors", "commit", "reset"], ph)), alive=alive, checksum=chk)))
'''

config = sys.argv[1]
envs = None
libs = sys.argv[2:]
if libs and libs[0].isdigit():
    envs = int(libs[0])
    libs = libs[1:]
for lib in libs:
    env = dict(os.environ, SMX_LIBRARY=os.path.abspath(lib))
    out = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, config=config, envs=envs)], env=env, capture_output=True, text=True)
    try:
        r = json.loads(out.stdout.strip().splitlines()[-1])
        print(f"{os.path.basename(lib):44s} tick {r['tick_ms']:.4f} ms  serial sum {sum(r['phases'].values()):.4f}  {r['phases']}  alive {r['alive']:.3f} chk {r['checksum']:.6f}", flush=True)
    except Exception:
        print(lib, "FAILED", out.stderr[-1500:], flush=True)
