import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from conftest import GOLDEN, SCENARIOS, MAPS
from smarts_amd.sumo_map import load_net
from smarts_amd.map_compiler import compile_map
from smarts_amd.missions import PlannedMission
from test_gpu_golden import _host, _sim_at_poses, differing_waypoint_rows

name = "4lane"
cm = compile_map(load_net(os.path.join(SCENARIOS, MAPS[name])))
g = np.load(os.path.join(GOLDEN, f"missions_{name}.npz"))
off = g["route_off"]
routes = [[str(r) for r in g["route_roads"][off[k]:off[k + 1]]] for k in range(int(g["n_routes"]))]
P, W = 8, 33
for k, roads in enumerate(routes):
    rows = np.flatnonzero(g["pose_route"] == k)
    res = {}
    for strategy in ("small", "smallstep", "large"):
        sim = _sim_at_poses(cm, g["poses"][rows], wp_paths=P, wp_len=W, wp_lookahead=32, launch_strategy=strategy.replace("step", ""))
        sim.set_missions([PlannedMission((0.0, 0.0), 0.0, (1e7, 1e7, 1.0), tuple(roads))])
        out = sim.reset()
        if strategy != "small":
            out = sim.step(torch.full((len(rows), 1), -1, dtype=torch.int8, device="cuda"))
        res[strategy] = {k2: _host(out[k2]).copy() for k2 in ("wp_count", "wp_pos", "wp_lane_id", "ego_pos")}
        res[strategy]["seed"] = _host(sim.seed_cache).copy() if hasattr(sim, "seed_cache") else None
        sim.close()
    for other in ("smallstep", "large"):
        a, b = res["small"], res[other]
        print(other, "differ", len([i for i in range(len(rows)) if not (np.array_equal(a["wp_count"][i], b["wp_count"][i]) and np.array_equal(a["wp_pos"][i], b["wp_pos"][i]))]))
    a, b = res["small"], res["large"]
    bad = [i for i in range(len(rows)) if not (np.array_equal(a["wp_count"][i], b["wp_count"][i]) and np.array_equal(a["wp_pos"][i], b["wp_pos"][i]))]
    print("route", k, roads, "differ", len(bad), "of", len(rows))
    for i in bad[:1]:
        print("  pose", rows[i], g["poses"][rows[i]], "ego small/large", a["ego_pos"][i, 0], b["ego_pos"][i, 0])
        for kk in range(3):
            print("   row", kk, "nonzero x small", int((a["wp_pos"][i, 0, kk, :, 0] != 0).sum()), "large", int((b["wp_pos"][i, 0, kk, :, 0] != 0).sum()),
                  "first diff idx", int(np.argmax(a["wp_pos"][i, 0, kk, :, 0] != b["wp_pos"][i, 0, kk, :, 0])))
        print("   counts small", a["wp_count"][i, 0], "large", b["wp_count"][i, 0])
        print("   lanes small", [cm.lane_ids[j] if j >= 0 else None for j in a["wp_lane_id"][i, 0, :, 0]])
        print("   lanes large", [cm.lane_ids[j] if j >= 0 else None for j in b["wp_lane_id"][i, 0, :, 0]])
        if a["seed"] is not None:
            print("   seeds small", a["seed"].reshape(9, -1)[:, i], "large", b["seed"].reshape(9, -1)[:, i])
