"""Developer: why vehicles reach k_waypoints_emit's slow list (library built with -DSMX_DEBUG_TIMING: python -m smarts_amd.build --prof).
    python tools/dev_slow_reasons.py [c4] [ticks]"""
import os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['SMX_LIBRARY'] = os.path.join(ROOT, 'smarts_amd', 'libsmarts_mi355x_prof.so')
import bench
from smarts_amd.sumo_map import load_net
from smarts_amd.map_compiler import compile_map
from smarts_amd.engine import BatchedSim, SimConfig, make_spawns
from smarts_amd import _native as nat
config = sys.argv[1] if len(sys.argv) > 1 else "c4"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 40
preset, scenario, cfg_kw = bench.workload_config(config)
E, N = cfg_kw["num_envs"], cfg_kw["num_vehicles"]
cm = compile_map(load_net(os.path.join(ROOT, 'smarts_amd/scenarios', scenario)))
sim = BatchedSim(cm, SimConfig(**cfg_kw), spawns=make_spawns(cm, E, N, episodes=4, seed=42)); lib = nat.load_library()
actions = torch.from_numpy(bench.action_stream(E, N, 42, 0)).cuda()
sim.reset()
for i in range(10): sim.step(actions[i % bench.ACTION_CYCLE])
torch.cuda.synchronize(); buf = (ctypes.c_ulonglong * 128)(); lib.smx_prof_read(buf, 1)
for i in range(T): sim.step(actions[(10 + i) % bench.ACTION_CYCLE])
torch.cuda.synchronize(); lib.smx_prof_read(buf, 1)
names = {50: 'teams on a road of more than four lanes', 51: 'teams with a branching inside the lookahead', 52: 'rows whose knot list was cut (nk > cap)',
         53: 'rows with more knots than a path lane holds', 54: 'rows the pool had no room for', 55: 'new vehicles', 56: 'live teams'}
for k in sorted(names):
    print(f'{names[k]:52s} {buf[k] / T:10.2f} per tick')
