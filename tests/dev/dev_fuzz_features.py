"""Developer: teacher-forced parity of the optional features (DAGM, IDM social traffic, float spaces) over seeds."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from smarts_amd.sumo_map import load_net
from smarts_amd.map_compiler import compile_map
from smarts_amd.engine import BatchedSim, SimConfig, make_spawns
import parity
def host(o):
    torch.cuda.synchronize(); return {k: v.cpu().numpy().reshape((-1,) + tuple(v.shape[2:])) for k, v in o.items() if k not in ('env_done', 'learner')}
total_bad = 0
for scn, E, agents, social, T in (('loop', 2, 4, 14, 50), ('intersections/4lane', 2, 4, 10, 40), ('minicity', 1, 6, 20, 25)):
    net = load_net(os.path.join(ROOT, 'smarts_amd/scenarios', scn)); cm = compile_map(net)
    for seed in range(int(sys.argv[1]), int(sys.argv[2])):
        N = agents + social
        cfg = SimConfig(num_envs=E, num_vehicles=N, num_social=social, social_model='idm', social_speed_factor=0.9 + 0.02 * (seed % 5),
                        neighbors=True, nb_radius=50.0, done_on_shoulder=False, dagm=True, dagm_width=32, dagm_height=32, dagm_resolution=50 / 32)
        spawns, where = make_spawns(cm, E, N, episodes=2, seed=seed, return_lanes=True)
        sim = BatchedSim(cm, cfg, spawns=spawns, social_spawns=where); ob = parity.OracleBatch(net, cm, cfg, spawns[0], where[0])
        bad = parity.compare(host(sim.reset()), ob.reset_observe(), where='reset ')
        rng = np.random.default_rng(seed)
        for t in range(T):
            acts = np.where(rng.random((E, N)) < 0.6, 0, rng.integers(1, 4, (E, N))).astype(np.int8)
            d = host(sim.step(torch.from_numpy(acts).cuda())); o = ob.step(acts)
            bad += parity.compare(d, o, tol64=1e-9, tol32=2e-5, where=f'{scn} seed{seed} t{t} ')
            if bad: break
            parity.sync_oracle_from_device(ob, sim)
        print(scn, 'seed', seed, 'ok' if not bad else bad[:3], flush=True)
        total_bad += len(bad); sim.close()
print('TOTAL BAD', total_bad)
