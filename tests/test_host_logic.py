"""Host-side product logic (no GPU): map compiler tables, controller gain constants, spawn
table, and the C-ABI surface of the built library."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

MAP_NAMES = ["loop", "4lane", "minicity"]


@pytest.mark.parametrize("name", MAP_NAMES)
def test_map_compiler_lanepoints_match_reference(name, compiled_maps):
    """The compiled lanepoint table equals the reference's LanePoints.from_sumo output
    (fixture from tests/golden/gen_golden.py) bit for bit, in the reference's order."""
    cm = compiled_maps(name)
    g = np.load(os.path.join(GOLDEN, f"lanepoints_{name}.npz"))
    n = len(g["x"])
    total = int(g["total"]) if "total" in g else n
    assert cm.n_lanepoints == total
    assert np.array_equal(cm.lp_x[:n], g["x"]) and np.array_equal(cm.lp_y[:n], g["y"])
    assert np.array_equal(cm.lp_heading[:n], g["heading"])
    assert np.array_equal(cm.lp_inferred[:n], g["inferred"])
    gl = list(g["lane_ids"])
    assert all(cm.lane_ids[cm.lp_lane[i]] == gl[g["lane"][i]] for i in range(n))
    assert np.array_equal(cm.lp_next_off[: n + 1], g["next_off"])
    assert np.array_equal(cm.lp_next_idx[: g["next_off"][n]], g["next_idx"])
    if "sum_x" in g:
        assert cm.lp_x.sum() == float(g["sum_x"]) and cm.lp_heading.sum() == float(g["sum_heading"])


@pytest.mark.parametrize("name", MAP_NAMES)
def test_grids_cover_every_item(name, compiled_maps):
    cm = compiled_maps(name)
    # every lanepoint sits in exactly the cell its coordinates hash to
    cx = np.floor((cm.lp_x - cm.lpg_origin[0]) / cm.lpg_cell).astype(int)
    cy = np.floor((cm.lp_y - cm.lpg_origin[1]) / cm.lpg_cell).astype(int)
    assert (cx >= 0).all() and (cx < cm.lpg_dims[0]).all() and (cy >= 0).all() and (cy < cm.lpg_dims[1]).all()
    cell = cy * cm.lpg_dims[0] + cx
    assert len(cm.lpg_idx) == cm.n_lanepoints
    for c in np.unique(cell)[:200]:
        members = set(cm.lpg_idx[cm.lpg_off[c]: cm.lpg_off[c + 1]].tolist())
        assert members == set(np.nonzero(cell == c)[0].tolist())
    # every segment is listed in every cell its bounding box touches
    assert set(cm.sg_idx.tolist()) == set(range(len(cm.seg_lane)))
    assert cm.max_fanout <= 15  # BranchState packs 4 bits per level


def test_lateral_gains_saturate_at_clip_bounds():
    """lane_following_controller.py:420-430: for the sedan and every Lane-space target speed the
    pole-placement gains fall outside the clip window, so the kernels use the bounds as constants."""
    from oracle.controller import lateral_gains

    for ts in (15, 12.5, 15.0):
        hg, lg = lateral_gains(ts, 3.68 / 2, 2356.0, 2681.95008628, 100000.0)
        assert (hg, lg) == (0.04, 3.4)
    assert lateral_gains(0, 3.68 / 2, 2356.0, 2681.95008628, 100000.0) == (0.01, 0.36)


def test_spawn_table(compiled_maps):
    from smarts_amd.engine import make_spawns

    cm = compiled_maps("loop")
    sp = make_spawns(cm, 3, 8, episodes=2, seed=42)
    assert sp.shape == (2, 24, 4)
    # env e uses PCG64(seed + e): shifting first_env reproduces the same rows (shard independence)
    sp2 = make_spawns(cm, 1, 8, episodes=2, seed=42, first_env=2)
    assert np.array_equal(sp[:, 16:24], sp2)
    # speed = lane speed limit, vehicles on one lane at least 8 m apart along it
    assert np.allclose(sp[..., 3], 16.67)
    d = np.linalg.norm(sp[0, :8, None, :2] - sp[0, None, :8, :2], axis=-1) + np.eye(8) * 100
    assert d.min() > 3.0


def test_library_exports_every_declared_symbol():
    """include/smx.h <-> libsmarts_mi355x.so: every declared entry point is exported."""
    from smarts_amd import _native as nat
    from smarts_amd import build

    lib_path = build.build()
    lib = ctypes.CDLL(lib_path)
    header = open(os.path.join(ROOT, "include", "smx.h")).read()
    declared = set(re.findall(r"\b(smx_[a-z_]+)\s*\(", header))
    assert {"smx_create", "smx_load_map", "smx_reset", "smx_step", "smx_destroy"} <= declared
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in smx.h but not exported"
    for sym in nat.EXPORTS:
        assert hasattr(lib, sym)
    lib.smx_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.smx_version()
    # the ctypes mirrors have the sizes the library was compiled with
    lib.smx_struct_size.restype = ctypes.c_uint64
    for which, mirror in enumerate((nat.SmxConfig, nat.SmxMapTables, nat.SmxState, nat.SmxSpawns, nat.SmxOutputs)):
        assert lib.smx_struct_size(which) == ctypes.sizeof(mirror), mirror.__name__
    nat.load_library(lib_path)  # runs the same check and binds the prototypes


def test_no_cpu_fallback(compiled_maps):
    """Without a GPU the product path refuses to run (it must not route through the oracle)."""
    import torch

    from smarts_amd import _native as nat
    from smarts_amd.engine import BatchedSim, SimConfig

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(nat.NativeLibraryError):
        BatchedSim(compiled_maps("loop"), SimConfig(num_envs=1, num_vehicles=2))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "smarts_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f


def test_ackermann_formula_reproduces_place_poles():
    """smx_vehicle.h lateral_gains_for_speed: the reference places poles with scipy
    (lane_following_controller.py:391-430); for this single-input system the gain row is unique and
    Ackermann's formula gives it.  Checked across the clip window of the heading gain."""
    import warnings

    from scipy import signal

    from oracle import controller as ctl
    from oracle import dynamics as dyn

    L, M, IZ, C = dyn.CHASSIS_LENGTH / 2, dyn.CHASSIS_MASS, dyn.CHASSIS_INERTIA_Z, dyn.ROAD_STIFFNESS
    assert (M, IZ, C) == (2356.0, 2681.95008628, 100000.0)  # the literals in smx_vehicle.h
    poles = np.array(ctl.DESIRED_POLES, dtype=float)

    def mats(v):
        A = np.array([[0, v, 0, v], [0, 0, 1, 0], [0, 0, -(2 * C * L ** 2) / (v * IZ), 0], [0, 0, -1, -2 * C / (M * v)]])
        B = np.array([[0], [0], [L * C / IZ], [C / (M * v)]])
        return A, B

    def ackermann(v):
        A, B = mats(v)
        Cm = np.hstack([B, A @ B, A @ A @ B, A @ A @ A @ B])
        phi = np.eye(4)
        for p in poles:
            phi = phi @ (A - p * np.eye(4))
        return np.linalg.solve(Cm.T, np.array([0, 0, 0, 1.0])) @ phi

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for v in (0.3, 1.0, 2.0, 2.02, 2.03, 2.04, 2.06, 3.0, 8.0, 15.0, 30.0):
            A, B = mats(v)
            K = signal.place_poles(A, B, poles, method="KNV0").gain_matrix[0]
            Ka = ackermann(v)
            assert abs(K[0] - Ka[0]) <= 1e-9 * abs(K[0]) and abs(K[1] - Ka[1]) <= 1e-9 * max(abs(K[1]), 1e-3)
            h, l = ctl.lateral_gains(v, L, M, IZ, C)
            assert l == 3.4 and h == np.clip(Ka[1], 0.02, 0.04) or abs(h - np.clip(Ka[1], 0.02, 0.04)) < 1e-11


def test_compiled_map_cache_round_trip(tmp_path):
    """scenario_build: the `scl scenario build` twin for the map (cli/studio.py:54-84) — cache hit
    returns identical tables, a changed source map invalidates it."""
    import dataclasses
    import shutil

    from smarts_amd.map_compiler import compile_map, pack_tables
    from smarts_amd.scenario_build import CACHE_NAME, build_scenario, load_compiled_map
    from smarts_amd.sumo_map import load_net

    src = os.path.join(ROOT, "smarts_amd", "scenarios", "loop")
    d = tmp_path / "loop"
    d.mkdir()
    shutil.copy(os.path.join(src, "map.smxnet.json.gz"), d)
    first = load_compiled_map(str(d))
    assert first.extras["from_cache"] is False and (d / CACHE_NAME).exists()
    again = load_compiled_map(str(d))
    assert again.extras["from_cache"] is True
    ref = compile_map(load_net(src))
    packed = pack_tables(ref)
    for f in dataclasses.fields(ref):
        if f.name != "extras":
            assert np.array_equal(np.asarray(getattr(ref, f.name)), np.asarray(getattr(again, f.name))), f.name
    for k, v in packed.items():
        assert np.array_equal(v, again.extras["packed"][k]) and v.dtype == again.extras["packed"][k].dtype
    # a different source map under the same name: the digest no longer matches
    shutil.copy(os.path.join(ROOT, "smarts_amd", "scenarios", "intersections", "4lane", "map.smxnet.json.gz"), d)
    other = load_compiled_map(str(d))
    assert other.extras["from_cache"] is False and other.n_lanes != ref.n_lanes
    assert build_scenario(str(d)).endswith(CACHE_NAME)


def test_oracle_drivable_area_raster_properties(nets):
    """oracle/sensors_extra.dagm: the reference's own check (test_observations.py:150-153: a vehicle on
    the road is on a drivable pixel), plus rotation with the ego heading and the grid's extent."""
    import math

    from oracle.dynamics import VehicleBody
    from oracle.road_network import ORoadNetwork
    from oracle.sensors_extra import dagm

    rmap = ORoadNetwork(nets("loop"))
    shape = nets("loop").all_lanes()[0].getShape(False)
    (x1, y1), (x2, y2) = shape[0], shape[1]
    mx, my = 0.5 * (x1 + x2), 0.5 * (y1 + y2)
    heading = math.atan2(y2 - y1, x2 - x1) - 0.5 * math.pi  # along the lane
    g = dagm(VehicleBody(mx, my, heading, 0.0), rmap.lane_bands(), 64, 64, 50 / 64)
    assert g.dtype == np.uint8 and g.shape == (64, 64) and set(np.unique(g)) == {0, 255}
    assert g[30:34, 30:34].min() == 255  # on the centre line: drivable all round
    # the lane runs "up" the image: the centre column is road where the lane is, columns far to the side are not
    assert g[:, 32].mean() > g[:, 2].mean()
    # same place, turned a quarter: the image turns with the vehicle
    q = dagm(VehicleBody(mx, my, heading + 0.5 * math.pi, 0.0), rmap.lane_bands(), 64, 64, 50 / 64)
    assert np.mean(np.rot90(g, -1) == q) > 0.97
    # far from any road: nothing
    assert dagm(VehicleBody(mx + 5000.0, my, 0.0, 0.0), rmap.lane_bands(), 32, 32, 1.0).max() == 0


def test_oracle_idm_follower_keeps_its_gap(nets):
    """oracle/sim.py::SocialBody.idm_speed: free road -> the desired speed; a standing leader ahead in the
    corridor -> the follower stops short of it; a vehicle beside the corridor is ignored."""
    from oracle.dynamics import VehicleBody
    from oracle.road_network import ORoadNetwork
    from oracle.sim import SocialBody

    rmap = ORoadNetwork(nets("loop"))
    lane = rmap.lane_by_id(nets("loop").all_lanes()[0].getID())
    f = SocialBody(0.0, 0.0, 0.0, 0.0, lane, 5.0, 7, 1.0)
    f.step(0.1)  # places it on its lane
    v0 = lane.speed_limit
    for _ in range(400):  # free road
        f.speed_cmd = f.idm_speed([], 0.1)
        f.u = f.speed_cmd
    assert f.u == pytest.approx(v0, rel=1e-3)
    # a standing leader 40 m ahead along the heading
    fx, fy = -np.sin(f.heading), np.cos(f.heading)
    leader = VehicleBody(f.x + 40.0 * fx, f.y + 40.0 * fy, f.heading, 0.0)
    beside = VehicleBody(f.x + 10.0 * fx + 3.2 * fy, f.y + 10.0 * fy - 3.2 * fx, f.heading, 0.0)
    x, y, travelled = f.x, f.y, 0.0
    for _ in range(600):
        f.x, f.y = x + travelled * fx, y + travelled * fy  # straight-line stand-in for the lane
        f.u = f.idm_speed([(0, leader), (1, beside)], 0.1)
        travelled += f.u * 0.1
    gap = 40.0 - travelled - 3.68
    assert f.u < 0.05 and 0.5 < gap < 4.0  # stopped, bumper gap about the minimum gap of 2.5 m


def test_bench_cpu_worker_leg_runs_without_a_gpu():
    """bench.py's all-cores CPU leg starts children like this one; they must not need torch or a GPU."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--cpu-worker", "1", "--cpu-seconds", "0.3"],
                         capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, out.stderr[-500:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["env_steps"] >= 1 and rec["seconds"] > 0


def test_c_abi_rejects_invalid_configs_before_touching_the_device():
    """smx_create validates the whole smx_config first (include/smx.h: 0 = OK, negative = error, message
    from smx_last_error): no GPU is needed to be told that a config cannot run."""
    import ctypes as C

    from smarts_amd import _native as nat

    lib = nat.load_library()

    def create(**over):
        c = nat.SmxConfig()
        c.num_envs, c.num_vehicles, c.dt = 2, 4, 0.1
        c.sensors = nat.SENSOR_WAYPOINTS | nat.SENSOR_NEIGHBORS
        c.wp_lookahead, c.wp_paths, c.wp_len, c.nb_max, c.nb_radius = 32, 4, 20, 10, 50.0
        for k, v in over.items():
            setattr(c, k, v)
        h = C.c_void_p(0xDEAD)
        rc = lib.smx_create(C.byref(c), 0, C.byref(h))
        # a failed create leaves no handle behind (nothing for the caller to remember to destroy); the
        # reason is read with a NULL handle
        assert rc < 0 and not h.value, (rc, h.value)
        return rc, lib.smx_last_error(None).decode()

    bad = [
        (dict(num_vehicles=65), "num_vehicles"), (dict(num_envs=0), "num_envs"), (dict(dt=0.0), "dt"),
        (dict(wp_len=34), "wp_len"), (dict(wp_lookahead=40), "lookahead"), (dict(wp_paths=65), "wp_paths"),
        (dict(via_max=33), "via_max"), (dict(num_social=4), "num_social"), (dict(action_space=9), "action_space"),
        (dict(social_model=7), "social_model"), (dict(nb_max=200), "nb_max"),
        (dict(sensors=nat.SENSOR_OGM, ogm_width=300, ogm_height=300, ogm_resolution=0.2), "ogm"),
        (dict(sensors=nat.SENSOR_OGM, ogm_width=64, ogm_height=64, ogm_resolution=0.0), "ogm"),
        (dict(sensors=nat.SENSOR_DAGM, dagm_width=10, dagm_height=10, dagm_resolution=1.0), "dagm"),
        (dict(sensors=nat.SENSOR_LIDAR, lidar_rays=0), "lidar"), (dict(alive_lists=5), "agents_alive"),
    ]
    for over, word in bad:
        rc, msg = create(**over)
        assert rc < 0 and word in msg, (over, rc, msg)
    assert lib.smx_create(None, 0, None) < 0


def _declared_buffers(E=3, N=4, **cfg_over):
    """An smx_config plus state / spawns / outputs structs whose pointers are fake (non-NULL integers, never
    dereferenced by smx_check_buffers) and whose declared extents are exactly what the configuration needs."""
    import ctypes as C

    from smarts_amd import _native as nat

    c = nat.SmxConfig()
    c.num_envs, c.num_vehicles, c.dt = E, N, 0.1
    c.sensors = nat.SENSOR_WAYPOINTS | nat.SENSOR_NEIGHBORS | nat.SENSOR_OGM | nat.SENSOR_LIDAR
    c.wp_lookahead, c.wp_paths, c.wp_len, c.nb_max, c.nb_radius = 32, 4, 20, 10, 50.0
    c.ogm_width, c.ogm_height, c.ogm_resolution, c.lidar_rays, c.lidar_max_distance = 64, 64, 0.5, 100, 20.0
    for k, v in cfg_over.items():
        setattr(c, k, v)
    T, PW, K, R = E * N, c.wp_paths * c.wp_len, c.nb_max, c.lidar_rays
    st, sp, out = nat.SmxState(), nat.SmxSpawns(), nat.SmxOutputs()
    state = dict(f64=(nat.S_COUNT * T, nat.DT_F64), flags=(T, nat.DT_I32), steps=(T, nat.DT_I32), env_ticks=(E, nat.DT_I32),
                 env_done_count=(E, nat.DT_I32), env_episode=(E, nat.DT_I32), driven_path=(T * 500, nat.DT_F64),
                 seed_cache=(nat.SEED_COUNT * T, nat.DT_I32), facts_i32=(nat.FACT_I_COUNT * T, nat.DT_I32),
                 facts_f64=(nat.FACT_F_COUNT * T, nat.DT_F64), env_reset_pending=(E, nat.DT_I32))
    for k, name in enumerate(nat.STATE_BUFFERS):
        setattr(st, name, 0x1000 + k)
        st.count[k], st.dtype[k] = state[name]
    sp.episodes, sp.pose, sp.pose_count = 2, 0x2000, 2 * T * 4
    outs = dict(ego_pos=(3 * T, nat.DT_F64), ego_f32=(nat.EGO_F32_COUNT * T, nat.DT_F32), ego_lane=(2 * T, nat.DT_I16),
                events=(9 * T, nat.DT_U8), reward=(T, nat.DT_F64), dist=(T, nat.DT_F64), done=(T, nat.DT_U8),
                active=(T, nat.DT_U8), env_done=(E, nat.DT_U8), learner=(2 * T, nat.DT_F32),
                wp_pos=(T * PW * 3, nat.DT_F64), wp_heading=(T * PW, nat.DT_F32), wp_lane_width=(T * PW, nat.DT_F32),
                wp_speed_limit=(T * PW, nat.DT_F32), wp_lane_index=(T * PW, nat.DT_I8), wp_lane_id=(T * PW, nat.DT_I16),
                wp_count=(T * (c.wp_paths + 1), nat.DT_U8), nb_pos=(T * K * 3, nat.DT_F64), nb_box=(T * K * 3, nat.DT_F32),
                nb_heading=(T * K, nat.DT_F32), nb_speed=(T * K, nat.DT_F32), nb_lane_index=(T * K, nat.DT_I8),
                nb_lane_id=(T * K, nat.DT_I16), nb_slot=(T * K, nat.DT_I8), nb_count=(T, nat.DT_U8),
                ogm=(T * 64 * 64, nat.DT_U8), lidar_hit=(T * R, nat.DT_U8), lidar_point=(T * R * 3, nat.DT_F64),
                collidees=(T, nat.DT_U64))
    for k, name in enumerate(nat.OUTPUT_FIELDS):
        if name in outs:
            setattr(out, name, 0x3000 + k)
            out.count[k], out.dtype[k] = outs[name]
    return c, st, sp, out


def _check(c, st, sp, out, has_vias=0):
    import ctypes as C

    from smarts_amd import _native as nat

    lib = nat.load_library()
    err = C.create_string_buffer(512)
    rc = lib.smx_check_buffers(C.byref(c), has_vias, C.byref(st), C.byref(sp), C.byref(out), err, 512)
    return rc, err.value.decode()


def test_entry_check_accepts_exact_extents_and_names_the_short_buffer():
    """pointers + element counts + dtype enum checked on entry.  A buffer one element short of
    what the configuration implies would be an out-of-bounds device write: it is refused, by name, without a
    device (smx_check_buffers is the check smx_reset / smx_step* run first)."""
    from smarts_amd import _native as nat

    c, st, sp, out = _declared_buffers()
    assert _check(c, st, sp, out) == (0, "")
    for names, struct in ((nat.STATE_BUFFERS, st), (nat.OUTPUT_FIELDS, out)):
        for k, name in enumerate(names):
            if not getattr(struct, name):
                continue
            struct.count[k] -= 1
            rc, msg = _check(c, st, sp, out)
            assert rc == -1 and name in msg and "elements declared" in msg, (name, rc, msg)
            struct.count[k] += 1
    sp.pose_count -= 1
    rc, msg = _check(c, st, sp, out)
    assert rc == -1 and "spawns.pose" in msg
    sp.pose_count += 1
    assert _check(c, st, sp, out)[0] == 0


def test_entry_check_dtype_null_and_optional_buffers():
    from smarts_amd import _native as nat

    c, st, sp, out = _declared_buffers()
    k = nat.OUTPUT_FIELDS.index("wp_heading")
    out.dtype[k] = nat.DT_F64  # a float64 tensor where the ABI writes float32
    rc, msg = _check(c, st, sp, out)
    assert rc == -1 and "wp_heading" in msg and "dtype" in msg
    out.dtype[k] = nat.DT_F32
    # a required buffer left NULL
    keep = out.ogm
    out.ogm = None
    rc, msg = _check(c, st, sp, out)
    assert rc == -1 and "out.ogm is NULL" in msg
    out.ogm = keep
    # optional ones may be NULL: the learner block, the collidee masks, the driven-path ring (unless not_moving is a
    # done criterion), the social spawn table (unless there are social vehicles)
    out.learner, out.collidees, st.driven_path = None, None, None
    assert _check(c, st, sp, out)[0] == 0
    c.done_criteria = nat.DONE_NOT_MOVING
    rc, msg = _check(c, st, sp, out)
    assert rc == -1 and "driven_path" in msg
    c.done_criteria = 0
    c.num_social = 1
    rc, msg = _check(c, st, sp, out)
    assert rc == -1 and "spawns.social" in msg
    c.num_social = 0
    # via rows are needed only once vias were given
    c.via_max = 4
    assert _check(c, st, sp, out, has_vias=0)[0] == 0
    rc, msg = _check(c, st, sp, out, has_vias=1)
    assert rc == -1 and "via_near" in msg
    # an empty spawn table
    sp.episodes = 0
    rc, msg = _check(c, st, sp, out)
    assert rc == -1 and "spawn table is empty" in msg


def test_release_library_has_no_debug_switches():
    """The timing switches (SMX_DEBUG_SKIP) exist only in the -DSMX_DEBUG_TIMING developer variant: the shipped
    library never reads the environment variable, so nothing can silently drop work from a timed tick."""
    from smarts_amd import build

    blob = open(build.build(), "rb").read()
    assert b"SMX_DEBUG_SKIP" not in blob
    src = open(os.path.join(ROOT, "smarts_amd", "csrc", "smx_kernels.hip")).read()
    assert src.count('getenv("SMX_DEBUG_SKIP")') == 1 and "#ifdef SMX_DEBUG_TIMING\n  if (const char* dbg = getenv" in src


def test_build_staleness_sees_every_header(tmp_path):
    """build.is_stale() derives its dependency list from csrc/* and include/*.h (round 1 listed the headers by
    hand and missed smx_scan.h: an edit to the scan code then reused a stale library)."""
    import time

    from smarts_amd import build

    deps = {os.path.basename(d) for d in build.dependencies()}
    assert {"smx_kernels.hip", "smx_scan.h", "smx_roadmap.h", "smx_vehicle.h", "smx_device.h", "smx.h", "build.py"} <= deps
    lib = build.build()
    assert not build.is_stale(lib)
    scan = os.path.join(ROOT, "smarts_amd", "csrc", "smx_scan.h")
    st = os.stat(scan)
    try:
        os.utime(scan, (time.time() + 5, time.time() + 5))
        assert build.is_stale(lib)
    finally:
        os.utime(scan, (st.st_atime, st.st_mtime))
    assert not build.is_stale(lib)
    assert not [f for f in os.listdir(os.path.dirname(lib)) if ".so.tmp." in f]  # the atomic-rename temporaries are gone
