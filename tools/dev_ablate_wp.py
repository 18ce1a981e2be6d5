"""Developer: sensors-phase time of the C4 tick with pieces of the waypoint kernels switched off
(SMX_DEBUG_SKIP bits; -DSMX_DEBUG_TIMING library only)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, torch, numpy as np
sys.path.insert(0, %r)
from smarts_amd.sumo_map import load_net
from smarts_amd.map_compiler import compile_map
from smarts_amd.engine import BatchedSim, SimConfig
cm = compile_map(load_net(os.path.join(%r, 'smarts_amd/scenarios/loop')))
E, N = 4096, 32
cfg = SimConfig(num_envs=E, num_vehicles=N, neighbors=True, nb_radius=50.0, auto_reset=True, ogm=True, ogm_width=64, ogm_height=64, ogm_resolution=50/64)
sim = BatchedSim(cm, cfg, spawn_episodes=2); sim.reset()
acts = torch.zeros((E, N), dtype=torch.int8, device='cuda')
for _ in range(20): sim.step(acts)
sim.set_timing(2)
for _ in range(40): sim.step(acts)
torch.cuda.synchronize()
print(' '.join('%%.3f' %% x for x in sim.read_phase_ms().mean(axis=0)))
''' % (ROOT, ROOT)
# variants are built beforehand: python -c "from smarts_amd import build; build.build_variant('_ab<mask>', ['SMX_ABLATE=<mask>'])"
for mask, name in [(0, 'full'), (12582912 | 134217728, '-eval -stores, fill loads of record 0'), (12582912 | 268435456, '-eval -stores, fill without accumulation'), (12582912, '-eval -stores'), (12582912 | 16777216, '-eval -stores -fill'), (12582912 | 16777216 | 33554432, '-eval -stores -fill -trip'), (67108864, '-whole tables kernel')]:
    lib = os.path.join(ROOT, 'smarts_amd', f'libsmarts_mi355x_ab{mask}.so')
    if not os.path.exists(lib):
        continue
    env = dict(os.environ, SMX_LIBRARY=lib)
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True)
    print(f'{name:28s} phases control scan ogm sensors commit reset: {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]}')
