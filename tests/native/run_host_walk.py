"""Child process of tests/test_host_walk.py (started with the sanitizer runtime preloaded): drives the
host-compiled device code over the reference-generated fixtures and prints one JSON line of mismatches."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from smarts_amd import _native as nat  # noqa: E402
from smarts_amd.map_compiler import compile_map, map_tables_struct  # noqa: E402
from smarts_amd.sumo_map import load_net  # noqa: E402

MAPS = {"loop": "loop", "4lane": "intersections/4lane", "minicity": "minicity"}
lib = C.CDLL(sys.argv[1])
dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
lib.host_waypoint_paths.argtypes = [C.POINTER(nat.SmxMapTables), C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int,
                                    ip, dp, dp, dp, dp, dp, ip]
lib.host_routed_waypoint_paths.argtypes = [C.POINTER(nat.SmxMapTables), C.POINTER(C.c_int16), C.POINTER(C.c_uint8), C.c_int, C.c_double,
                                           C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, ip, dp, dp, dp, dp, dp, ip]
lib.host_nearest_lane.argtypes = [C.POINTER(nat.SmxMapTables), C.c_double, C.c_double, C.c_double, dp, ip]
out = {}
for name in sys.argv[2:]:
    cm = compile_map(load_net(os.path.join(ROOT, "smarts_amd", "scenarios", MAPS[name])))
    tables, keep = map_tables_struct(cm)
    golden = os.path.join(ROOT, "tests", "golden")
    for lookahead in (16, 32):
        w = np.load(os.path.join(golden, f"waypoints_{name}_empty_route_{lookahead}.npz"))
        lane_no = np.array([cm.lane_ids.index(str(l)) for l in w["lane_ids"]])
        MAXP, ST = 16, 36
        n = np.zeros(MAXP, np.int32)
        x, y, h, wd, sp = (np.zeros(MAXP * ST) for _ in range(5))
        ln = np.zeros(MAXP * ST, np.int32)
        bad, worst = [], 0.0
        for i, (px, py, ph) in enumerate(w["poses"]):
            cnt = lib.host_waypoint_paths(C.byref(tables), px, py, ph, lookahead, MAXP, ST, n.ctypes.data_as(ip),
                                          *(a.ctypes.data_as(dp) for a in (x, y, h, wd, sp)), ln.ctypes.data_as(ip))
            p0, p1 = w["path_off"][i], w["path_off"][i + 1]
            ok = cnt == p1 - p0
            for k in range(min(cnt, p1 - p0, MAXP)):
                a, b = w["wp_off"][p0 + k], w["wp_off"][p0 + k + 1]
                ok = ok and n[k] == b - a
                if not ok:
                    break
                sl, me = slice(a, b), slice(k * ST, k * ST + b - a)
                ok = ok and np.array_equal(ln[me], lane_no[w["lane"][sl]])
                err = max(np.abs(x[me] - w["x"][sl]).max(), np.abs(y[me] - w["y"][sl]).max(), np.abs(h[me] - w["heading"][sl]).max(),
                          np.abs(wd[me] - w["width"][sl]).max(), np.abs(sp[me] - w["speed"][sl]).max())
                worst = max(worst, float(err))
                ok = ok and err <= 1e-9  # (the device accumulates the projection in another order than numpy: last-ulp differences)
            if not ok:
                bad.append(i)
        out[f"waypoints_{name}_{lookahead}"] = dict(differing=bad, worst=worst, poses=len(w["poses"]))
    # ---- fixed routes (missions_<map>.npz): smx_set_missions' per-slot tables, built here by the same rule
    g = np.load(os.path.join(golden, f"missions_{name}.npz"))
    lane_no = np.array([cm.lane_ids.index(str(l)) for l in g["lane_ids"]])
    off = g["route_off"]
    w = {k[len("wp32_"):]: g[k] for k in g.files if k.startswith("wp32_")}
    bad, worst = [], 0.0
    for r in range(int(g["n_routes"])):
        roads = [cm.road_ids.index(str(x)) for x in g["route_roads"][off[r]:off[r + 1]]]
        pos = np.full(len(cm.road_ids), -1, np.int16)
        for i_, rd in enumerate(roads):
            if pos[rd] < 0:
                pos[rd] = i_
        ok_lane = np.zeros(len(cm.lane_ids), np.uint8)
        for ln_ in range(len(cm.lane_ids)):
            rd = cm.lane_road[ln_]
            good = pos[rd] >= 0
            if good and rd != roads[-1]:
                good = any(pos[cm.lane_road[o_]] >= 0 for o_ in cm.lane_out_idx[cm.lane_out_off[ln_]:cm.lane_out_off[ln_ + 1]])
            ok_lane[ln_] = 1 if good else 0
        for i in np.flatnonzero(g["pose_route"] == r):
            px, py, ph = g["poses"][i]
            cnt = lib.host_routed_waypoint_paths(C.byref(tables), pos.ctypes.data_as(C.POINTER(C.c_int16)),
                                                 ok_lane.ctypes.data_as(C.POINTER(C.c_uint8)), int(roads[-1]), px, py, ph, 32, MAXP, ST,
                                                 n.ctypes.data_as(ip), *(a.ctypes.data_as(dp) for a in (x, y, h, wd, sp)),
                                                 ln.ctypes.data_as(ip))
            p0, p1 = w["path_off"][i], w["path_off"][i + 1]
            ok = cnt == p1 - p0
            for k in range(min(cnt, p1 - p0, MAXP)):
                a, b = w["wp_off"][p0 + k], w["wp_off"][p0 + k + 1]
                ok = ok and n[k] == b - a
                if not ok:
                    break
                sl, me = slice(a, b), slice(k * ST, k * ST + b - a)
                ok = ok and np.array_equal(ln[me], lane_no[w["lane"][sl]])
                err = max(np.abs(x[me] - w["x"][sl]).max(), np.abs(y[me] - w["y"][sl]).max(), np.abs(h[me] - w["heading"][sl]).max(),
                          np.abs(wd[me] - w["width"][sl]).max(), np.abs(sp[me] - w["speed"][sl]).max())
                worst = max(worst, float(err))
                ok = ok and err <= 1e-9
            if not ok:
                bad.append(int(i))
    out[f"routed_{name}"] = dict(differing=bad, worst=worst, poses=len(g["poses"]))
    g = np.load(os.path.join(golden, f"nearest_{name}.npz"))
    lane_no = np.array([cm.lane_ids.index(str(l)) for l in g["lane_ids"]] + [-1])
    bad = []
    d, on = C.c_double(), C.c_int()
    for i, (px, py, ph) in enumerate(g["poses"]):
        lane = lib.host_nearest_lane(C.byref(tables), px, py, 10.0, C.byref(d), C.byref(on))
        if lane != lane_no[g["nearest"][i]] or (lane >= 0 and abs(d.value - g["dist"][i]) > 1e-9) or bool(on.value) != bool(g["on_road"][i]):
            bad.append(i)
    out[f"nearest_{name}"] = dict(differing=bad, poses=len(g["poses"]))
print(json.dumps(out))
