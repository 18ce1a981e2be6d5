"""Developer: run the same scenario in two sims; report the first tick/field where they differ."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smarts_amd.sumo_map import load_net
from smarts_amd.map_compiler import compile_map
from smarts_amd.engine import BatchedSim, SimConfig, make_spawns
from smarts_amd import _native as nat
scn, E, N, T, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
cm = compile_map(load_net(os.path.join(ROOT, 'smarts_amd/scenarios', scn)))
cfg = SimConfig(num_envs=E, num_vehicles=N, neighbors=True, nb_radius=50.0)
spawns = make_spawns(cm, E, N, episodes=2, seed=seed)
sims = [BatchedSim(cm, cfg, spawns=spawns) for _ in range(2)]
for s in sims: s.reset()
rng = np.random.default_rng(seed)
def snap(s):
    torch.cuda.synchronize()
    d = {('out', k): v.cpu().numpy() for k, v in s.out.items()}
    d[('st', 'state')] = s.state.cpu().numpy(); d[('st', 'flags')] = s.flags.cpu().numpy(); d[('st', 'seeds')] = s.seed_cache.cpu().numpy()
    d[('st', 'facts_i')] = s.facts_i32.cpu().numpy(); d[('st', 'facts_f')] = s.facts_f64.cpu().numpy()
    return d
nd = 0
for t in range(T):
    acts = torch.from_numpy(np.where(rng.random((E, N)) < 0.8, 0, rng.integers(1, 4, (E, N))).astype(np.int8)).cuda()
    for s in sims: s.step(acts)
    a, b = snap(sims[0]), snap(sims[1])
    for k in a:
        if not np.array_equal(a[k], b[k], equal_nan=True):
            idx = np.argwhere(a[k] != b[k])[0]
            print(f't{t} {k} differs at {idx.tolist()}: {a[k][tuple(idx)]} vs {b[k][tuple(idx)]}  (count {int((a[k] != b[k]).sum())})'); nd += 1
    if nd: break
print('done', 'NONDETERMINISTIC' if nd else f'deterministic over {T} ticks')
