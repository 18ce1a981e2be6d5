"""Developer: teacher-forced run; on the first mismatching tick print controller internals of the oracle."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from smarts_amd.sumo_map import load_net
from smarts_amd.map_compiler import compile_map
from smarts_amd.engine import BatchedSim, SimConfig, make_spawns
from smarts_amd import _native as nat
import parity, copy, ctypes
pass
from oracle import controller as ctl
scn, E, N, T, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
net = load_net(os.path.join(ROOT, 'smarts_amd/scenarios', scn)); cm = compile_map(net)
cfg = SimConfig(num_envs=E, num_vehicles=N, neighbors=True, nb_radius=50.0)
spawns = make_spawns(cm, E, N, episodes=2, seed=seed)
sim = BatchedSim(cm, cfg, spawns=spawns); ob = parity.OracleBatch(net, cm, cfg, spawns[0])
def host(o):
    torch.cuda.synchronize(); return {k: v.cpu().numpy().reshape((-1,) + tuple(v.shape[2:])) for k, v in o.items() if k != 'env_done'}
host(sim.reset()); ob.reset_observe()
rng = np.random.default_rng(seed)
for t in range(T):
    acts = np.where(rng.random((E, N)) < 0.8, 0, rng.integers(1, 4, (E, N))).astype(np.int8)
    if t % 7 == 3: acts[0, 0] = -1
    # snapshot oracle pre-step state for diagnostics
    pre = [(copy.deepcopy(ag.body), copy.deepcopy(ag.ctrl)) for env in ob.envs for ag in env.agents]
    torch.cuda.synchronize(); st_pre = sim.state.cpu().numpy().reshape(nat.S_COUNT, -1).copy(); fl_pre = sim.flags.cpu().numpy().reshape(-1).copy()
    d = host(sim.step(torch.from_numpy(acts).cuda())); o = ob.step(acts)
    bad = parity.compare(d, o, tol64=1e-9, tol32=2e-5, where=f't{t} ')
    if bad:
        print('\n'.join(bad[:4]))
        err = np.abs(d['ego_pos'] - o['ego_pos']).max(axis=1); g = int(np.argmax(err)); e, i = divmod(g, N)
        body, ctrl = pre[g]; a = int(acts[e, i]); ts, lc = ctl.LANE_ACTIONS[ctl.LANE_ACTION_NAMES[a]]
        rm_ = ob.road_map
        paths = rm_.waypoint_paths(body.position, body.heading, lookahead=16, route=())
        cur = ctl.find_current_lane(paths, body.position); want = int(np.clip(cur + lc, 0, len(paths) - 1))
        print('veh', g, 'action', a, 'n_paths', len(paths), 'lens', [len(p) for p in paths], 'cur', int(cur), 'want', want, 'lane ids', [p[0].lane_id for p in paths])
        print('pre pose', body.x, body.y, body.heading, 'u', body.u, 'ctrl steer', ctrl.steering_state, 'mcl', ctrl.min_curvature_location)
        st = sim.state.cpu().numpy().reshape(nat.S_COUNT, -1)
        ag = ob.envs[e].agents[i]
        print('post oracle steer %.6f thr %.6f latint %.6f | dev steer %.6f thr %.6f latint %.6f' % (ag.ctrl.steering_state, ag.ctrl.throttle_state, ag.ctrl.lateral_integral_error, st[nat.S['STEER'], g], st[nat.S['THROTTLE'], g], st[nat.S['LAT_INT'], g]))
        print('DEV mcl(lax,lay)', st[nat.S['MCL_X'], g], st[nat.S['MCL_Y'], g], 'code', st[nat.S['SPD_INT'], g], 'lah', st[nat.S['SPD_ERR'], g])
        for p in paths: print('oracle wps', [(round(float(w.pos[0]),6), round(float(w.pos[1]),6), round(float(w.heading),6)) for w in p[:6]])
        print('DEV pre flags', fl_pre[g], 'mcl set', bool(fl_pre[g] & nat.F_MCL_SET), 'mcl', st_pre[nat.S['MCL_X'], g], st_pre[nat.S['MCL_Y'], g], 'steer', st_pre[nat.S['STEER'], g], 'latint', st_pre[nat.S['LAT_INT'], g], 'spd_int', st_pre[nat.S['SPD_INT'], g])
        seeds = sim.seed_cache.cpu().numpy().reshape(9, -1)[:, g]; print('dev seeds now', seeds)
        break
    parity.sync_oracle_from_device(ob, sim)
else:
    print('no mismatch in', T)
