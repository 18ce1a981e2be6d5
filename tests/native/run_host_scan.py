"""Child process of tests/test_host_scan.py (sanitizer runtime preloaded): drives the host-compiled scan over random
drives on the three maps and prints one JSON line.  For every step the SEEDED search (carry = the answers at the
previous pose) must give exactly what the unseeded search gives at the new pose."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from smarts_amd import _native as nat  # noqa: E402
from smarts_amd.engine import make_spawns  # noqa: E402
from smarts_amd.map_compiler import compile_map, map_tables_struct  # noqa: E402
from smarts_amd.sumo_map import load_net  # noqa: E402

MAPS = {"loop": "loop", "4lane": "intersections/4lane", "minicity": "minicity"}
lib = C.CDLL(sys.argv[1])
steps = int(sys.argv[2])
dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
T = C.POINTER(nat.SmxMapTables)
lib.host_scan_facts.argtypes = [T, C.c_double, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, dp, ip]
lib.host_scan_seeds.argtypes = [T, C.c_double, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, ip, ip, dp]


def facts(tables, x, y, h, thr_max, carry=None, mode=1):
    out = (C.c_double * 3)()
    fl = C.c_int()
    qx, qy, pd = carry if carry else (0.0, 0.0, 0.0)
    lane = lib.host_scan_facts(tables, x, y, h, mode if carry else 0, qx, qy, pd, thr_max, out, C.byref(fl))
    return lane, out[0], fl.value & 255, out[1], out[2], bool(fl.value & 256)


def seeds(tables, x, y, h, carry=None, mode=1):
    oi = (C.c_int * 20)()
    od = (C.c_double * 10)()
    if carry:
        qx, qy, d10, d1, road, lanes, starts = carry
        st = (C.c_int * 4)(*starts)
        lib.host_scan_seeds(tables, x, y, h, mode, qx, qy, d10, d1, road, lanes, st, oi, od)
    else:
        st = (C.c_int * 4)(-1, -1, -1, -1)
        lib.host_scan_seeds(tables, x, y, h, 0, 0.0, 0.0, -1.0, -1.0, -1, 0, st, oi, od)
    return list(oi), list(od)


out = {}
for name in sys.argv[3:]:
    cm = compile_map(load_net(os.path.join(ROOT, "smarts_amd", "scenarios", MAPS[name])))
    tables, keep = map_tables_struct(cm)
    tp = C.byref(tables)
    thr_max = 0.5 * float(cm.lane_width.max()) + 0.1
    rng = np.random.default_rng(7)
    spawns = make_spawns(cm, 6, 16, episodes=1, seed=11)[0]
    bad_f, bad_s, bad_h, bad_f1, bad_s1, guessed, served_f, served_s, n = [], [], [], [], [], 0, 0, 0, 0
    for v, (x, y, h, _) in enumerate(spawns):
        f_prev = facts(tp, x, y, h, thr_max)
        si_prev, sd_prev = seeds(tp, x, y, h)
        for k in range(steps):
            # a drive: mostly forward 0..2 m with some sideways drift and heading noise; now and then a jump
            step = rng.uniform(0.0, 2.0) if rng.random() > 0.03 else rng.uniform(2.0, 30.0)
            h2 = h + rng.normal(0.0, 0.08)
            x2 = x - np.sin(h2) * step + rng.normal(0.0, 0.15)
            y2 = y + np.cos(h2) * step + rng.normal(0.0, 0.15)
            f_ref = facts(tp, x2, y2, h2, thr_max)
            f_carry = (x, y, f_prev[1]) if f_prev[0] >= 0 else None
            f_sd = facts(tp, x2, y2, h2, thr_max, carry=f_carry)
            if f_ref[:3] != f_sd[:3]:
                bad_f.append((v, k))
            if f_ref[0] >= 0 and f_ref[3] != f_ref[4]:
                bad_h.append((v, k))
            f_one = facts(tp, x2, y2, h2, thr_max, carry=f_carry, mode=2)  # the one-lane two-pass form
            if f_ref[:5] != f_one[:5]:
                bad_f1.append((v, k))
            served_f += 1 if f_one[5] else 0
            si_ref, sd_ref = seeds(tp, x2, y2, h2)
            carry = (x, y, sd_prev[9] if si_prev[9] >= 0 else -1.0, sd_prev[0] if si_prev[0] >= 0 else -1.0, si_prev[10],
                     si_prev[14], si_prev[15:19])
            si_sd, sd_sd = seeds(tp, x2, y2, h2, carry=carry)
            if si_ref[:19] != si_sd[:19] or sd_ref != sd_sd:
                bad_s.append((v, k))
            guessed += 1 if si_sd[19] >= 0 else 0
            si_one, sd_one = seeds(tp, x2, y2, h2, carry=carry, mode=2)  # the one-lane form without the ten-nearest list
            if si_one[19] == 1:
                served_s += 1
                if si_one[10:19] != si_ref[10:19] or sd_one[0] != sd_ref[0]:
                    bad_s1.append((v, k))
            n += 1
            f_prev, si_prev, sd_prev = f_ref, si_ref, sd_ref
            x, y = x2, y2
            h = f_ref[3] if (f_ref[0] >= 0 and rng.random() < 0.7) else h2
            if f_ref[0] < 0:  # left the map: back to the spawn
                x, y, h = spawns[v][:3]
                f_prev = facts(tp, x, y, h, thr_max)
                si_prev, sd_prev = seeds(tp, x, y, h)
    out[name] = dict(steps=n, facts_differ=bad_f[:5], seeds_differ=bad_s[:5], heading_forms_differ=bad_h[:5], guessed=guessed,
                     one_lane_facts_differ=bad_f1[:5], one_lane_seeds_differ=bad_s1[:5], one_lane_facts_served=served_f,
                     one_lane_seeds_served=served_s)
print(json.dumps(out))
