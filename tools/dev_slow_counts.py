"""Developer: lengths of the large form's slow lists tick by tick (smx_debug_slow_counts).   python tools/dev_slow_counts.py [c4] [ticks]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from smarts_amd.engine import BatchedSim, SimConfig, make_spawns
from smarts_amd.map_compiler import compile_map
from smarts_amd.sumo_map import load_net
config = sys.argv[1] if len(sys.argv) > 1 else "c4"
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 40
preset, scenario, cfg_kw = bench.workload_config(config)
E, N = cfg_kw["num_envs"], cfg_kw["num_vehicles"]
cm = compile_map(load_net(os.path.join(ROOT, "smarts_amd", "scenarios", scenario)))
sim = BatchedSim(cm, SimConfig(**cfg_kw), spawns=make_spawns(cm, E, N, episodes=4, seed=42))
actions = torch.from_numpy(bench.action_stream(E, N, 42, 0)).cuda()
sim.reset()
buf = (C.c_int32 * 4)()
sim.lib.smx_debug_slow_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
for i in range(ticks):
    sim.step(actions[i % bench.ACTION_CYCLE])
    if i < 6 or i % 8 == 0:
        sim.lib.smx_debug_slow_counts(sim.handle, buf)
        alive = int(sim.out["active"].sum().item())
        print(f"tick {i:3d}: alive {alive:6d}  slow facts {buf[0]:5d}  seeds {buf[1]:5d}  control {buf[2]:5d}  rows {buf[3]:5d}")
