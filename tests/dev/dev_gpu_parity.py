"""Developer script: device vs oracle parity over a rollout + a quick timing (run through gpurun)."""
import sys, time, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from smarts_amd.sumo_map import load_net
from smarts_amd.map_compiler import compile_map
from smarts_amd.engine import BatchedSim, SimConfig, make_spawns
from smarts_amd import build
import parity

def host(o):
    torch.cuda.synchronize()
    return {k: v.cpu().numpy().reshape((-1,) + tuple(v.shape[2:])) for k, v in o.items() if k != 'env_done'}

def run(scn, E, N, T, seed):
    net = load_net(os.path.join(ROOT, 'smarts_amd/scenarios', scn)); cm = compile_map(net)
    cfg = SimConfig(num_envs=E, num_vehicles=N, neighbors=True, nb_radius=50.0)
    spawns = make_spawns(cm, E, N, episodes=1, seed=seed)
    sim = BatchedSim(cm, cfg, spawns=spawns)
    t0 = time.time(); ob = parity.OracleBatch(net, cm, cfg, spawns[0]); 
    d = host(sim.reset()); o = ob.reset_observe()
    bad = parity.compare(d, o, where='reset ')
    print(scn, 'reset mismatches:', bad[:5])
    rng = np.random.default_rng(seed); nbad = 0
    for t in range(T):
        acts = np.where(rng.random((E, N)) < 0.8, 0, rng.integers(1, 4, (E, N))).astype(np.int8)
        d = host(sim.step(torch.from_numpy(acts).cuda())); o = ob.step(acts)
        bad = parity.compare(d, o, tol64=1e-5, tol32=5e-4, where=f't{t} ')
        if bad:
            nbad += 1; print('\n'.join(bad[:6]))
            if nbad > 3: break
    print(scn, f'E={E} N={N} T={T}: ticks with mismatch {nbad}; still active {int(d["active"].sum())}/{E*N}; oracle time {time.time()-t0:.1f}s')
    sim.close()

def bench(scn, E, N, steps=50):
    net = load_net(os.path.join(ROOT, 'smarts_amd/scenarios', scn)); cm = compile_map(net)
    cfg = SimConfig(num_envs=E, num_vehicles=N, neighbors=True, nb_radius=50.0, auto_reset=True)
    t0=time.time(); sim = BatchedSim(cm, cfg, spawn_episodes=2); print('setup', time.time()-t0)
    sim.reset(); acts = torch.zeros((E, N), dtype=torch.int8, device='cuda')
    for _ in range(10): sim.step(acts)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(steps): sim.step(acts)
    torch.cuda.synchronize(); dt = (time.time() - t0) / steps
    print(f'{scn} E={E} N={N}: {dt*1e3:.3f} ms/tick -> {E/dt:.0f} env-steps/s, {E*N/dt:.0f} agent-steps/s; active {int(sim.out["active"].sum())}')
    sim.close()

if __name__ == '__main__':
    build.build()
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if what in ('all', 'parity'):
        run('loop', 6, 8, 150, 1)
        run('intersections/4lane', 4, 16, 100, 2)
        run('minicity', 2, 16, 60, 3)
    if what in ('all', 'bench'):
        bench('loop', 1024, 8)
        bench('loop', 4096, 32, steps=20)
