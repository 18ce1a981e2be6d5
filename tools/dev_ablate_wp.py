"""Developer: per-phase times of a BASELINE configuration's tick with pieces switched off at COMPILE time
(-DSMX_ABLATE=<mask>: no stamps, no run-time switches — what is left runs as the shipped code does).

    python tools/dev_ablate_wp.py c4 0 1 1048576 2097152

builds smarts_amd/libsmarts_mi355x_ab<mask>.so for every mask (here, on the CPU box, before gpurun) when called
with --build, and times them on the GPU otherwise.  Bits: smx_kernels.hip, SMX_SKIP(...)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = [a for a in sys.argv[1:] if a != "--build"]
config, masks = args[0], [int(m) for m in args[1:]]
if "--build" in sys.argv:
    from smarts_amd import build
    for m in masks:
        print(build.build_variant(f"_ab{m}", [f"SMX_ABLATE={m}"]))
    sys.exit(0)
code = r'''
import os, sys, torch, numpy as np
sys.path.insert(0, %r)
import bench
from smarts_amd.sumo_map import load_net
from smarts_amd.map_compiler import compile_map
from smarts_amd.engine import BatchedSim, SimConfig, make_spawns
preset, scenario, cfg_kw = bench.workload_config(%r)
E, N = cfg_kw["num_envs"], cfg_kw["num_vehicles"]
cm = compile_map(load_net(os.path.join(%r, "smarts_amd", "scenarios", scenario)))
sim = BatchedSim(cm, SimConfig(**cfg_kw), spawns=make_spawns(cm, E, N, episodes=4, seed=42))
actions = torch.from_numpy(bench.action_stream(E, N, 42, 0)).cuda()
sim.reset()
for i in range(20): sim.step(actions[i %% 64])
sim.set_timing(2)
for i in range(40): sim.step(actions[(20 + i) %% 64])
torch.cuda.synchronize()
print(' '.join('%%.3f' %% x for x in sim.read_phase_ms().mean(axis=0)))
''' % (ROOT, config, ROOT)
for m in masks:
    lib = os.path.join(ROOT, 'smarts_amd', f'libsmarts_mi355x_ab{m}.so')
    if not os.path.exists(lib):
        print(m, 'not built'); continue
    out = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, SMX_LIBRARY=lib), capture_output=True, text=True)
    print(f'ablate {m:>10d}: control scan ogm sensors commit reset = {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]}')
