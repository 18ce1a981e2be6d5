import os, sys, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from conftest import GOLDEN, SCENARIOS, MAPS
from smarts_amd.sumo_map import load_net
from smarts_amd.map_compiler import compile_map
from smarts_amd.missions import PlannedMission
from test_gpu_golden import _host, _sim_at_poses

name = "4lane"
cm = compile_map(load_net(os.path.join(SCENARIOS, MAPS[name])))
g = np.load(os.path.join(GOLDEN, f"missions_{name}.npz"))
off = g["route_off"]
k = 4
roads = [str(r) for r in g["route_roads"][off[k]:off[k + 1]]]
rows = np.flatnonzero(g["pose_route"] == k)
for strategy in ("small", "large"):
    sim = _sim_at_poses(cm, g["poses"][rows], wp_paths=8, wp_len=33, wp_lookahead=32, launch_strategy=strategy)
    sim.set_missions([PlannedMission((0.0, 0.0), 0.0, (1e7, 1e7, 1.0), tuple(roads))])
    sim.reset()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    sim.lib.smx_prof_read(buf, 1)
    out = sim.step(torch.full((len(rows), 1), -1, dtype=torch.int8, device="cuda"))
    torch.cuda.synchronize()
    sim.lib.smx_prof_read(buf, 1)
    print(strategy, "step prof[28:32]", list(buf)[28:32], "n_lanes", len(cm.lane_ids), "count row1", _host(out["wp_count"])[:, 0, 1].tolist()[:12])
    sim.close()
