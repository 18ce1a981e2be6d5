"""Developer: run a scenario with the bounds-checked build (SMX_LIBRARY=..._dbg.so) and report the first violation."""
import os, sys, ctypes, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['SMX_LIBRARY'] = os.path.join(ROOT, 'smarts_amd', 'libsmarts_mi355x_dbg.so')
from smarts_amd.sumo_map import load_net
from smarts_amd.map_compiler import compile_map
from smarts_amd.engine import BatchedSim, SimConfig, make_spawns
from smarts_amd import _native as nat
scn, E, N, T, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
cm = compile_map(load_net(os.path.join(ROOT, 'smarts_amd/scenarios', scn)))
cfg = SimConfig(num_envs=E, num_vehicles=N, neighbors=True, nb_radius=50.0)
sim = BatchedSim(cm, cfg, spawns=make_spawns(cm, E, N, episodes=2, seed=seed))
lib = nat.load_library()
def chk(tag):
    torch.cuda.synchronize(); s = ctypes.c_int(); v = ctypes.c_longlong()
    lib.smx_debug_read(ctypes.byref(s), ctypes.byref(v))
    if s.value: print(tag, 'BOUNDS VIOLATION site', s.value, 'value', v.value); sys.exit(0)
sim.reset(); chk('reset')
rng = np.random.default_rng(seed)
for t in range(T):
    acts = np.where(rng.random((E, N)) < 0.8, 0, rng.integers(1, 4, (E, N))).astype(np.int8)
    if t % 7 == 3: acts[0, 0] = -1
    sim.step(torch.from_numpy(acts).cuda()); chk(f't{t}')
print('no violation in', T, 'ticks; active', int(sim.out['active'].sum()))
