"""Developer: per-phase wave-clock breakdown of the tick kernel (build with -DSMX_DEBUG_TIMING)."""
import os, sys, ctypes, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['SMX_LIBRARY'] = os.path.join(ROOT, 'smarts_amd', 'libsmarts_mi355x_prof.so')
from smarts_amd.sumo_map import load_net
from smarts_amd.map_compiler import compile_map
from smarts_amd.engine import BatchedSim, SimConfig
from smarts_amd import _native as nat
E, N = int(sys.argv[1]), int(sys.argv[2]); scn = sys.argv[3] if len(sys.argv) > 3 else 'loop'
extra = dict(ogm=True, ogm_width=64, ogm_height=64, ogm_resolution=50 / 64) if 'ogm' in sys.argv[4:] else {}
if 'large' in sys.argv[4:]: extra['launch_strategy'] = 'large'
cm = compile_map(load_net(os.path.join(ROOT, 'smarts_amd/scenarios', scn)))
cfg = SimConfig(num_envs=E, num_vehicles=N, neighbors=True, nb_radius=50.0, auto_reset=True, **extra)
sim = BatchedSim(cm, cfg, spawn_episodes=2); lib = nat.load_library()
sim.reset(); acts = torch.zeros((E, N), dtype=torch.int8, device='cuda')
for _ in range(20): sim.step(acts)
torch.cuda.synchronize(); buf = (ctypes.c_ulonglong * 128)(); lib.smx_prof_read(buf, 1)
T = 50
for _ in range(T): sim.step(acts)
torch.cuda.synchronize(); lib.smx_prof_read(buf, 1)
names = {2: 'wp staged: pass B (interp + copy)', 24: 'wp staged: copy A', 25: 'wp staged: interp B', 26: 'wp staged: copy B', 
         9: 'seeds: heading terms of the 10 nearest', 0: 'wp: loads+seeds', 1: 'wp: paths (walk+emit | tables: walk)', 3: 'wp: total (lane-0 waves)', 4: 'esp pass1 (all callers)', 5: 'esp pass2 (all callers) + tables: serial tail', 6: 'observe: total', 7: 'observe: loads+barrier', 8: 'observe: collide+ego+nb',
         10: 'scan: road facts (8 lanes/veh)', 11: 'scan: lane heading', 12: 'scan: nearest10', 13: 'scan: path seeds', 14: 'scan: total', 21: 'seeds: pick_closest (+junction case)', 22: 'seeds: road -> lanes loads', 23: 'seeds: closest lanepoint per lane',
         15: 'control: loads', 16: 'control: path walk+synth', 17: 'control: reduce+shuffle', 18: 'control: law (lane 0)', 19: 'control: physics', 20: 'control: total (lane-0 waves)'}
lanes = {2: 4, 24: 4, 25: 4, 26: 4, 9: 8, 0: 4, 1: 4, 3: 4, 4: 4, 5: 4, 6: 1, 7: 1, 8: 1, 10: 8, 11: 8, 12: 8, 13: 8, 14: 8, 21: 8, 22: 8, 23: 8, 15: 4, 16: 4, 17: 4, 18: 4, 19: 4, 20: 4}
for k in sorted(names):
    waves = (E * N * lanes[k] + 63) // 64
    print(f'{names[k]:36s} {buf[k] / T / waves / 100.0:10.2f} us/wave (100 MHz clock)')
