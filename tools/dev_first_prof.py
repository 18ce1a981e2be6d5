"""Developer: where the reset pass (k_first) spends a restarted env group's time (library built with -DSMX_DEBUG_TIMING:
python -m smarts_amd.build --prof).   python tools/dev_first_prof.py [c5] [warm ticks] [ticks]"""
import os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['SMX_LIBRARY'] = os.path.join(ROOT, 'smarts_amd', 'libsmarts_mi355x_prof.so')
import bench
from smarts_amd.sumo_map import load_net
from smarts_amd.map_compiler import compile_map
from smarts_amd.engine import BatchedSim, SimConfig, make_spawns
from smarts_amd import _native as nat
config = sys.argv[1] if len(sys.argv) > 1 else "c5"
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 150
T = int(sys.argv[3]) if len(sys.argv) > 3 else 100
preset, scenario, cfg_kw = bench.workload_config(config)
E, N = cfg_kw["num_envs"], cfg_kw["num_vehicles"]
cm = compile_map(load_net(os.path.join(ROOT, 'smarts_amd/scenarios', scenario)))
sim = BatchedSim(cm, SimConfig(**cfg_kw), spawns=make_spawns(cm, E, N, episodes=4, seed=42)); lib = nat.load_library()
actions = torch.from_numpy(bench.action_stream(E, N, 42, 0)).cuda()
sim.reset()
for i in range(warm): sim.step(actions[i % bench.ACTION_CYCLE])
torch.cuda.synchronize(); buf = (ctypes.c_ulonglong * 128)(); lib.smx_prof_read(buf, 1)
for i in range(T): sim.step(actions[(warm + i) % bench.ACTION_CYCLE])
torch.cuda.synchronize(); lib.smx_prof_read(buf, 1)
names = {57: 'scan round (both halves, or the seeds halves)', 58: 'waypoint rows (beside them the facts halves left over)', 59: 'observe', 60: 'commit', 61: 'scan .. commit'}
print(f"{config}: k_first workgroups that found new vehicles, ticks {warm}-{warm + T} ({sim.launch_form()})")
for k in sorted(names):
    if buf[k + 64]:
        print(f'{names[k]:36s} {buf[k] / buf[k + 64] / 100.0:9.2f} us   ({buf[k + 64] / T:.1f} workgroups per tick)')
