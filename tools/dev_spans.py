"""Developer: per-wavefront spans of the large form's kernels — mean and longest (library built with -DSMX_DEBUG_TIMING:
python -m smarts_amd.build --prof).  A kernel ends with its slowest wavefront.   python tools/dev_spans.py [c4] [serial|forked]"""
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
serial = (sys.argv[2] if len(sys.argv) > 2 else "serial") == "serial"
preset, scenario, cfg_kw = bench.workload_config(config)
E, N = cfg_kw["num_envs"], cfg_kw["num_vehicles"]
cm = compile_map(load_net(os.path.join(ROOT, 'smarts_amd/scenarios', scenario)))
sim = BatchedSim(cm, SimConfig(**cfg_kw), spawns=make_spawns(cm, E, N, episodes=4, seed=42)); lib = nat.load_library()
actions = torch.from_numpy(bench.action_stream(E, N, 42, 0)).cuda()
sim.reset()
for i in range(10): sim.step(actions[i % bench.ACTION_CYCLE])
if serial: sim.set_timing(2)  # one kernel at a time
names = ['k_control_fast', 'k_scan_fast<1> (seeds)', 'k_scan_fast<0> (facts)', 'k_wp_walk', 'k_waypoints_emit', 'k_observe', 'k_ogm_env']
import numpy as np
KN, WN = 8, 16384
buf = (ctypes.c_uint * (KN * WN))()
waves = [2048, 2048, 2048, 8192, 8192, 2048, 16384]
T = 30
rows = [[] for _ in names]
for i in range(T):
    sim.step(actions[(10 + i) % bench.ACTION_CYCLE])
    torch.cuda.synchronize(); lib.smx_span_read(buf)
    arr = np.frombuffer(buf, dtype=np.uint32).reshape(KN, WN) / 100.0
    for k in range(len(names)):
        rows[k].append(arr[k, :waves[k]].copy())
print(f"{config}, {'serial' if serial else 'forked'} ticks 10-{10 + T}: wavefront spans (us), per tick, averaged over the ticks")
for k, nm in enumerate(names):
    a = np.stack(rows[k]); a = np.where(a > 0, a, np.nan)
    q = lambda p: float(np.nanmean(np.nanpercentile(a, p, axis=1)))
    print(f'{nm:26s} mean {float(np.nanmean(a)):7.2f}  p50 {q(50):7.2f}  p90 {q(90):7.2f}  p99 {q(99):7.2f}  longest {q(100):7.2f}')
