"""Developer: long auto-reset run at BASELINE configs[1] size; checks finiteness and episode turnover."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smarts_amd.sumo_map import load_net
from smarts_amd.map_compiler import compile_map
from smarts_amd.engine import BatchedSim, SimConfig
scn = sys.argv[1] if len(sys.argv) > 1 else 'loop'; T = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
E, N = 1024, 8
cm = compile_map(load_net(os.path.join(ROOT, 'smarts_amd/scenarios', scn)))
sim = BatchedSim(cm, SimConfig(num_envs=E, num_vehicles=N, neighbors=True, nb_radius=50.0, auto_reset=True, max_episode_steps=400), spawn_episodes=8)
sim.reset(); rng = np.random.default_rng(0)
acts = torch.from_numpy(np.where(rng.random((64, E, N)) < 0.8, 0, rng.integers(1, 4, (64, E, N))).astype(np.int8)).cuda()
t0 = time.time(); done_total = 0; env_done_total = 0
for t in range(T):
    o = sim.step(acts[t % 64])
    if t % 250 == 249:
        torch.cuda.synchronize()
        assert torch.isfinite(o['ego_pos']).all() and torch.isfinite(o['wp_pos']).all() and torch.isfinite(sim.state).all(), t
        ev = o['events'].float().mean(dim=(0, 1)).cpu().numpy().round(4)
        print(t + 1, 'active', float(o['active'].float().mean()), 'episodes min/max', int(sim.env_episode.min()), int(sim.env_episode.max()), 'events', ev, flush=True)
torch.cuda.synchronize(); print('ok', T, 'ticks in', round(time.time() - t0, 2), 's')
