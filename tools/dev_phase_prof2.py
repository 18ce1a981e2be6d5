"""Developer: wave-clock stamps of the large form's kernels (library built with -DSMX_DEBUG_TIMING: python -m smarts_amd.build --prof).
    python tools/dev_phase_prof2.py [c4]"""
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
preset, scenario, cfg_kw = bench.workload_config(config)
E, N = cfg_kw["num_envs"], cfg_kw["num_vehicles"]
cm = compile_map(load_net(os.path.join(ROOT, 'smarts_amd/scenarios', scenario)))
sim = BatchedSim(cm, SimConfig(**cfg_kw), spawns=make_spawns(cm, E, N, episodes=4, seed=42)); lib = nat.load_library()
actions = torch.from_numpy(bench.action_stream(E, N, 42, 0)).cuda()
sim.reset()
for i in range(10): sim.step(actions[i % bench.ACTION_CYCLE])
sim.set_timing(2)  # serial: one kernel at a time
torch.cuda.synchronize(); buf = (ctypes.c_ulonglong * 128)(); lib.smx_prof_read(buf, 1)
T = 30
for i in range(T): sim.step(actions[(10 + i) % bench.ACTION_CYCLE])
torch.cuda.synchronize(); lib.smx_prof_read(buf, 1)
V = E * N
names = {32: ('emit: prologue loads', V * 4), 33: ('emit: knot walk', V * 4), 34: ('emit: prefix sum, book, pool writes', V * 4), 35: ('emit: slots loop', V * 4), 36: ('emit: trip meter', V * 4), 37: ('emit: total', V * 4),
         38: ('emit: a round of 4 slots, arithmetic', V * 4 * 5), 39: ('emit: a round of 4 slots, fixups + stores', V * 4 * 5),
         40: ('facts: setup + pass 1', V), 41: ('facts: pass 2', V), 42: ('facts: heading candidates', V), 43: ('facts: two lane positions', V), 44: ('facts: trig', V)}
for k in sorted(names):
    nm, lanes = names[k]
    if buf[k + 64]:
        print(f'{nm:44s} {buf[k] / buf[k + 64] / 100.0:10.2f} us per report  ({buf[k + 64] / T:.0f} reports per tick)')
