"""Developer: run a BASELINE configuration with per-kernel timing on (smx_set_timing(2): every kernel of the
tick on the caller's stream, no side streams) — under `rocprofv3 --kernel-trace --stats` the kernel averages
are then those of kernels that ran alone.   python tools/dev_kernel_times.py [c4|c5|c3|c2] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from smarts_amd.engine import BatchedSim, SimConfig, make_spawns
from smarts_amd.map_compiler import compile_map
from smarts_amd.sumo_map import load_net
config = sys.argv[1] if len(sys.argv) > 1 else "c4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
preset, scenario, cfg_kw = bench.workload_config(config)
E, N = cfg_kw["num_envs"], cfg_kw["num_vehicles"]
cm = compile_map(load_net(os.path.join(ROOT, "smarts_amd", "scenarios", scenario)))
sim = BatchedSim(cm, SimConfig(**cfg_kw), spawns=make_spawns(cm, E, N, episodes=4, seed=42))
actions = torch.from_numpy(bench.action_stream(E, N, 42, 0)).cuda()
sim.reset()
for i in range(10):
    sim.step(actions[i % bench.ACTION_CYCLE])
sim.set_timing(2)
for i in range(steps):
    sim.step(actions[(10 + i) % bench.ACTION_CYCLE])
torch.cuda.synchronize()
print(dict(zip(["control", "scan", "ogm", "sensors", "commit", "reset"], sim.read_phase_ms().mean(axis=0).round(4))))
