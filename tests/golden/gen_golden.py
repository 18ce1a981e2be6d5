#!/usr/bin/env python3
"""Generate golden vectors by running the reference's OWN Python on our parsed maps.

Runs only in the build container (needs ``/root/reference``); the GPU box and the
test-suite consume the committed ``tests/golden/*.npz`` / ``*.json`` outputs.

How the reference is driven:

* ``smarts`` is imported from ``/root/reference`` through a namespace shim
  (``smarts/__init__.py`` insists on an installed distribution), with
  ``sys.dont_write_bytecode`` so nothing is written into the read-only tree.
* Third-party modules that are absent here (``sumolib``, ``shapely``, ``trimesh``,
  ``pybullet`` …) are replaced by *name-only* stub modules so that the
  reference's modules import.  The only stubbed *functions* that are ever
  called are ``sumolib.geomhelper.positionAtShapeOffset`` /
  ``polygonOffsetWithMinimumDistanceToPoint`` / ``distance``, which delegate to
  the reference's own in-tree twins (``smarts/core/utils/math.py:293-390``), and
  ``pybullet.getQuaternionFromEuler`` (closed-form, for lidar rays).
* The reference's ``SumoRoadNetwork`` is instantiated around our parsed
  network object (``smarts_amd.sumo_map.SumoNet`` exposes the sumolib accessor
  names), so ``LanePoints.from_sumo``, ``waypoint_paths``,
  ``_equally_spaced_path``, ``nearest_lanes`` etc. are the reference's code.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py
"""
import importlib
import importlib.machinery
import json
import logging
import math
import os
import random
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("SMARTS_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, "tests", "golden")

import numpy as np  # noqa: E402


class _Anything:
    """Attribute sink used for names that are imported but never executed."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []

    def _mod_getattr(n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Anything

    m.__getattr__ = _mod_getattr  # type: ignore
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    if "." in name:
        parent, child = name.rsplit(".", 1)
        if parent in sys.modules:
            setattr(sys.modules[parent], child, m)
    return m


def install_reference():
    smarts = types.ModuleType("smarts")
    smarts.__path__ = [os.path.join(REF, "smarts")]
    smarts.__spec__ = importlib.machinery.ModuleSpec("smarts", None, is_package=True)
    sys.modules["smarts"] = smarts
    sys.path.insert(0, REF)  # for `envision`

    import functools

    _stub("cached_property", cached_property=functools.cached_property)
    for name in [
        "shapely", "shapely.geometry", "shapely.geometry.base", "shapely.affinity", "shapely.ops",
        "trimesh", "trimesh.scene", "trimesh.exchange", "trimesh.exchange.gltf", "trimesh.visual",
        "sh", "yattag", "tableprint", "websocket", "rtree", "gym", "gym.spaces",
        "pybullet_utils", "pybullet_utils.bullet_client",
        "sumolib", "sumolib.net", "sumolib.net.edge", "sumolib.net.lane", "sumolib.geomhelper",
        "sumo", "sumo.tools", "sumo.tools.sumolib", "sumo.tools.traci",
        "traci", "traci.constants", "traci.exceptions",
    ]:
        _stub(name)

    def _quat_from_euler(rpy):
        # pybullet.getQuaternionFromEuler (x, y, z, w), ZYX convention
        r, p, y = rpy
        cr, sr = math.cos(r * 0.5), math.sin(r * 0.5)
        cp, sp = math.cos(p * 0.5), math.sin(p * 0.5)
        cy, sy = math.cos(y * 0.5), math.sin(y * 0.5)
        return (
            sr * cp * cy - cr * sp * sy,
            cr * sp * cy + sr * cp * sy,
            cr * cp * sy - sr * sp * cy,
            cr * cp * cy + sr * sp * sy,
        )

    _stub("pybullet", getQuaternionFromEuler=_quat_from_euler, MAX_RAY_INTERSECTION_BATCH_SIZE=16384,
          DIRECT=2, GUI=1, TORQUE_CONTROL=1, POSITION_CONTROL=2, URDF_USE_IMPLICIT_CYLINDER=128,
          URDF_ENABLE_CACHED_GRAPHICS_SHAPES=1024,
          __all__=["getQuaternionFromEuler", "MAX_RAY_INTERSECTION_BATCH_SIZE", "DIRECT", "GUI", "TORQUE_CONTROL",
                   "POSITION_CONTROL", "URDF_USE_IMPLICIT_CYLINDER", "URDF_ENABLE_CACHED_GRAPHICS_SHAPES"])

    sys.modules["sumo"].SUMO_HOME = "/nonexistent-sumo-home"
    sys.modules["sumo.tools"].traci = sys.modules["sumo.tools.traci"]
    rmath = importlib.import_module("smarts.core.utils.math")
    gh = sys.modules["sumolib.geomhelper"]
    gh.positionAtShapeOffset = rmath.position_at_shape_offset
    gh.distance = rmath.euclidean_distance
    gh.polygonOffsetWithMinimumDistanceToPoint = (
        lambda point, shape, perpendicular=False: rmath.polygon_offset_with_minimum_distance_to_point(point, shape)
    )
    gh.distancePointToPolygon = rmath.distance_point_to_polygon
    sys.modules["sumolib"].geomhelper = gh
    sys.modules["sumo.tools.sumolib"].geomhelper = gh
    sys.modules["sumo.tools"].sumolib = sys.modules["sumolib"]
    return rmath


def make_reference_road_network(net, spacing=1.0):
    """The reference's SumoRoadNetwork around our parsed net (no sumolib)."""
    from smarts.core.lanepoints import LanePoints
    from smarts.core.sumo_road_network import SumoRoadNetwork

    # sumolib.Net API used by the reference that is not a plain accessor
    def getNeighboringLanes(x, y, r=0.1, includeJunctions=True, allowFallback=True):
        return net.neighboring_lanes(x, y, r, includeJunctions)

    net.getNeighboringLanes = getNeighboringLanes

    rn = SumoRoadNetwork.__new__(SumoRoadNetwork)
    rn._log = logging.getLogger("ref")
    rn._graph = net
    rn._net_file = net.source
    rn._map_spec = types.SimpleNamespace(lanepoint_spacing=spacing, default_lane_width=None,
                                         shift_to_origin=True, source=net.source)
    rn._default_lane_width = 3.2
    rn._surfaces = {}
    rn._lanes = {}
    rn._roads = {}
    rn._waypoints_cache = SumoRoadNetwork._WaypointsCache()
    rn._lanepoints = LanePoints.from_sumo(rn, spacing=spacing)
    return rn


def dump_lanepoints(rn):
    lps = rn._lanepoints._linked_lanepoints
    index = {id(l): i for i, l in enumerate(lps)}
    lanes = sorted({l.lp.lane.lane_id for l in lps})
    lane_no = {lid: i for i, lid in enumerate(lanes)}
    nxt_off = [0]
    nxt = []
    for l in lps:
        nxt += [index[id(n)] for n in l.nexts]
        nxt_off.append(len(nxt))
    return dict(
        x=np.array([l.lp.pose.position[0] for l in lps]),
        y=np.array([l.lp.pose.position[1] for l in lps]),
        heading=np.array([float(l.lp.pose.heading) for l in lps]),
        inferred=np.array([l.is_inferred for l in lps], dtype=np.uint8),
        lane=np.array([lane_no[l.lp.lane.lane_id] for l in lps], dtype=np.int32),
        lane_ids=np.array(lanes),
        next_off=np.array(nxt_off, dtype=np.int32),
        next_idx=np.array(nxt, dtype=np.int32),
    )


def sample_poses(net, rng, n, lateral=2.5, heading_noise=0.6, far_fraction=0.05):
    """Random poses near lanes (plus a few far off-road ones)."""
    from smarts_amd.sumo_map import polyline_point_at

    lanes = net.all_lanes()
    poses = []
    for i in range(n):
        lane = lanes[rng.integers(len(lanes))]
        shape = np.asarray(lane.shape, dtype=np.float64)
        seg = np.sqrt(((shape[1:] - shape[:-1]) ** 2).sum(axis=1))
        total = float(seg.sum())
        s = rng.uniform(0.0, total)
        x, y = polyline_point_at(shape, s)
        x2, y2 = polyline_point_at(shape, min(s + 0.5, total))
        if (x2, y2) == (x, y):
            x0, y0 = polyline_point_at(shape, max(s - 0.5, 0.0))
            dxv, dyv = x - x0, y - y0
        else:
            dxv, dyv = x2 - x, y2 - y
        h = math.atan2(dyv, dxv) - math.pi / 2
        lat = rng.normal(0.0, lateral / 2.5)
        if rng.random() < far_fraction:
            lat = rng.uniform(-30, 30)
        x += -math.sin(h + math.pi / 2) * lat
        y += math.cos(h + math.pi / 2) * lat
        h += rng.normal(0.0, heading_noise) if rng.random() < 0.8 else rng.uniform(-math.pi, math.pi)
        h = (h + math.pi) % (2 * math.pi) - math.pi
        poses.append((float(x), float(y), float(h)))
    return poses


def dump_waypoint_paths(rn, poses, lookahead, route_kind):
    """Run the reference's waypoint_paths for every pose; flatten the ragged result."""
    from smarts.core.coordinates import Heading, Pose
    from smarts.core.utils.math import fast_quaternion_from_angle

    lane_ids = sorted(l.getID() for l in rn._graph.all_lanes())
    lane_no = {lid: i for i, lid in enumerate(lane_ids)}
    rec = dict(path_off=[0], wp_off=[0], x=[], y=[], heading=[], lane=[], lane_index=[], width=[], speed=[])
    for (x, y, h) in poses:
        pose = Pose(position=np.array([x, y, 0.0]), orientation=fast_quaternion_from_angle(Heading(h)),
                    heading_=Heading(h))
        route = rn.empty_route() if route_kind == "empty_route" else None
        paths = rn.waypoint_paths(pose, lookahead=lookahead, route=route)
        for p in paths:
            for wp in p:
                rec["x"].append(float(wp.pos[0]))
                rec["y"].append(float(wp.pos[1]))
                rec["heading"].append(float(wp.heading))
                rec["lane"].append(lane_no[wp.lane_id])
                rec["lane_index"].append(int(wp.lane_index))
                rec["width"].append(float(wp.lane_width))
                rec["speed"].append(float(wp.speed_limit))
            rec["wp_off"].append(len(rec["x"]))
        rec["path_off"].append(len(rec["wp_off"]) - 1)
    out = {k: np.array(v) for k, v in rec.items()}
    out["lane_ids"] = np.array(lane_ids)
    out["poses"] = np.array(poses)
    out["lookahead"] = np.array(lookahead)
    return out


def dump_nearest(rn, poses):
    from smarts.core.coordinates import Point

    lane_ids = sorted(l.getID() for l in rn._graph.all_lanes())
    lane_no = {lid: i for i, lid in enumerate(lane_ids)}
    nearest = []
    dist = []
    on_road = []
    for (x, y, h) in poses:
        pt = Point(x, y, 0.0)
        nl = rn.nearest_lanes(pt)
        nearest.append(lane_no[nl[0][0].lane_id] if nl else -1)
        dist.append(nl[0][1] if nl else -1.0)
        on_road.append(rn.road_with_point(pt) is not None)
    return dict(poses=np.array(poses), nearest=np.array(nearest, dtype=np.int32), dist=np.array(dist),
                on_road=np.array(on_road, dtype=np.uint8), lane_ids=np.array(lane_ids))


def dump_controller(rn, net, rng, n):
    """Reference LaneFollowingController.perform_lane_following on mock vehicles
    (the reference's own tests drive sensors the same way: test_sensors.py:39-99)."""
    from smarts.core.chassis import AckermannChassis
    from smarts.core.controllers.lane_following_controller import (
        LaneFollowingController,
        LaneFollowingControllerState,
    )
    from smarts.core.coordinates import Heading, Point, Pose
    from smarts.core.utils.math import fast_quaternion_from_angle

    lane_ids = sorted(l.getID() for l in net.all_lanes())
    lane_no = {lid: i for i, lid in enumerate(lane_ids)}
    sim = types.SimpleNamespace(road_map=rn, last_dt=0.1, road_stiffness=100000.0)
    sensor_state = types.SimpleNamespace(plan=types.SimpleNamespace(route=rn.empty_route(), road_map=rn))
    poses = sample_poses(net, rng, n, lateral=1.5, heading_noise=0.15, far_fraction=0.0)
    actions = [(15, 0), (0, 0), (12.5, 1), (12.5, -1)]
    cols = {k: [] for k in [
        "x", "y", "z", "heading", "speed", "lat_speed", "long_speed", "yaw_z", "target_speed", "lane_change",
        "in_lat_int", "in_spd_int", "in_steer", "in_thr", "in_spd_err", "in_mcl_x", "in_mcl_y", "in_mcl_set",
        "in_target_lane",
        "throttle", "brake", "steering",
        "out_lat_int", "out_spd_int", "out_steer", "out_thr", "out_spd_err", "out_mcl_x", "out_mcl_y", "out_mcl_set",
        "out_hgain", "out_lgain", "out_target_lane",
    ]}
    for (x, y, h) in poses:
        speed = float(rng.uniform(0.0, 22.0)) if rng.random() < 0.9 else 0.0
        lat = float(rng.normal(0, 0.3))
        yawz = float(rng.normal(0, 0.2))
        z = 0.01265
        ts, lc = actions[rng.integers(4)] if rng.random() < 0.7 else actions[0]
        hd = Heading(h)
        pose = Pose(position=np.array([x, y, z]), orientation=fast_quaternion_from_angle(hd), heading_=hd)
        chassis = AckermannChassis.__new__(AckermannChassis)
        lng = math.sqrt(max(speed * speed - lat * lat, 0.0))
        chassis.__dict__["longitudinal_lateral_speed"] = (lng, lat)
        chassis.__dict__["velocity_vectors"] = (np.array([lng, lat, 0.0]), np.array([0.0, 0.0, yawz]))
        chassis.__dict__["mass_and_inertia"] = (2356.0, 2681.95008628)
        captured = {}

        def control(throttle=0, brake=0, steering=0, captured=captured):
            captured.update(throttle=float(throttle), brake=float(brake), steering=float(steering))

        vehicle = types.SimpleNamespace(
            chassis=chassis, pose=pose, position=pose.position, heading=hd, speed=speed, length=3.68,
            max_steering_wheel=12.56 / 17.4, control=control,
        )
        tl = rn.nearest_lane(Point(x, y, z))
        st = LaneFollowingControllerState(tl.lane_id)
        if rng.random() < 0.8:
            st.lateral_integral_error = float(rng.normal(0, 0.2))
            st.integral_speed_error = float(rng.normal(0, 2.0))
            st.steering_state = float(np.clip(rng.normal(0, 0.3), -1, 1))
            st.throttle_state = float(rng.uniform(0, 1))
            st.speed_error = float(rng.normal(0, 1.0))
        if rng.random() < 0.5:
            st.min_curvature_location = (x + float(rng.normal(0, 2)), y + float(rng.normal(0, 2)))
        rec_in = dict(
            in_lat_int=st.lateral_integral_error, in_spd_int=st.integral_speed_error, in_steer=st.steering_state,
            in_thr=st.throttle_state, in_spd_err=st.speed_error,
            in_mcl_set=int(st.min_curvature_location != (None, None)),
            in_mcl_x=st.min_curvature_location[0] or 0.0, in_mcl_y=st.min_curvature_location[1] or 0.0,
            in_target_lane=lane_no[st.target_lane_id],
        )
        try:
            LaneFollowingController.perform_lane_following(
                sim, "a", vehicle, st, sensor_state, target_speed=ts, lane_change=lc)
        except AssertionError:
            continue
        for k, v in rec_in.items():
            cols[k].append(v)
        for k, v in dict(x=x, y=y, z=z, heading=h, speed=speed, lat_speed=lat, long_speed=lng, yaw_z=yawz,
                         target_speed=ts, lane_change=lc).items():
            cols[k].append(v)
        for k in ("throttle", "brake", "steering"):
            cols[k].append(captured[k])
        cols["out_lat_int"].append(float(st.lateral_integral_error))
        cols["out_spd_int"].append(float(st.integral_speed_error))
        cols["out_steer"].append(float(st.steering_state))
        cols["out_thr"].append(float(st.throttle_state))
        cols["out_spd_err"].append(float(st.speed_error))
        cols["out_mcl_set"].append(int(st.min_curvature_location != (None, None)))
        cols["out_mcl_x"].append(st.min_curvature_location[0] or 0.0)
        cols["out_mcl_y"].append(st.min_curvature_location[1] or 0.0)
        cols["out_hgain"].append(float(st.heading_error_gain))
        cols["out_lgain"].append(float(st.lateral_error_gain))
        cols["out_target_lane"].append(lane_no[st.target_lane_id])
    out = {k: np.array(v) for k, v in cols.items()}
    out["lane_ids"] = np.array(lane_ids)
    return out


def dump_lidar_rays():
    """Reference Lidar._compute_rays (lidar.py:89-113) for BasicLidar and the 100-ray planar
    sensor of BASELINE config 5."""
    from smarts.core.lidar import Lidar
    from smarts.core.lidar_sensor_params import BasicLidar, SensorParams

    planar = SensorParams(start_angle=0, end_angle=2 * np.pi, laser_angles=np.array([0.0]),
                          angle_resolution=2 * np.pi / 100.5, max_distance=20, noise_mu=0, noise_sigma=0.078)
    out = {}
    for name, params in (("basic", BasicLidar), ("planar100", planar)):
        origin = np.array([3.0, -7.0, 1.5])
        lidar = Lidar(origin, params, None)
        rays = lidar._compute_rays()
        out[name] = np.array([d - origin for (_, d) in rays])
        out[name + "_origin_ok"] = np.array(all((o == origin).all() for (o, _) in rays))
    return out


SCENARIOS = {
    "loop": "scenarios/loop",
    "4lane": "scenarios/intersections/4lane",
    "minicity": "scenarios/minicity",
}


def dump_std_obs():
    """Reference ``lane_ttc`` (custom_observations.py:148-280) and ``FormatObs._std_*``
    (format_obs.py:401-603) on observations of an oracle rollout (loop, 1 env x 8 agents,
    waypoints + neighbours).  The reference objects are filled field by field from the dense rows
    (tests/parity.pack), which are stored as the fixture's input."""
    for name in ("gym.envs", "gym.envs.registration", "gym.wrappers"):
        _stub(name)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import parity
    from smarts.core.coordinates import Dimensions as RDimensions
    from smarts.core.coordinates import Heading as RHeading
    from smarts.core.events import Events as REvents
    from smarts.core.road_map import Waypoint as RWaypoint
    from smarts.core.sensors import EgoVehicleObservation as REgo
    from smarts.core.sensors import Observation as RObservation
    from smarts.core.sensors import VehicleObservation as RVehicle
    from smarts.core.sensors import Vias as RVias
    from smarts.env import custom_observations as rco
    from smarts.env.wrappers import format_obs as rfo

    from smarts_amd.engine import SimConfig, make_spawns
    from smarts_amd.env.observations import ObservationBuilder
    from smarts_amd.map_compiler import compile_map
    from smarts_amd.sumo_map import load_net

    net = load_net(os.path.join(REPO, "smarts_amd", "scenarios", "loop"))
    cm = compile_map(net)
    E, N = 1, 8
    cfg = SimConfig(num_envs=E, num_vehicles=N, neighbors=True, nb_radius=50.0)
    spawns = make_spawns(cm, E, N, episodes=1, seed=7)
    ob = parity.OracleBatch(net, cm, cfg, spawns[0])
    agent_ids = [f"agent_{i}" for i in range(N)]
    builder = ObservationBuilder(cm.lane_ids, [cm.road_ids[r] for r in cm.lane_road], agent_ids, waypoints=True,
                                 neighbors=True, accelerometer=True, dt=0.1)

    def to_ref(o):
        e = o.ego_vehicle_state
        ego = REgo(id=e.id, position=e.position, bounding_box=RDimensions(*e.bounding_box.as_lwh),
                   heading=RHeading(float(e.heading)), speed=e.speed, steering=e.steering, yaw_rate=e.yaw_rate,
                   road_id=e.road_id, lane_id=e.lane_id, lane_index=e.lane_index, mission=None,
                   linear_velocity=e.linear_velocity, angular_velocity=e.angular_velocity,
                   linear_acceleration=e.linear_acceleration, angular_acceleration=e.angular_acceleration,
                   linear_jerk=e.linear_jerk, angular_jerk=e.angular_jerk)
        nbs = [RVehicle(id=v.id, position=v.position, bounding_box=RDimensions(*v.bounding_box.as_lwh),
                        heading=RHeading(float(v.heading)), speed=v.speed, road_id=v.road_id, lane_id=v.lane_id,
                        lane_index=v.lane_index) for v in o.neighborhood_vehicle_states]
        paths = [[RWaypoint(pos=w.pos, heading=RHeading(float(w.heading)), lane_id=w.lane_id, lane_width=w.lane_width,
                            speed_limit=w.speed_limit, lane_index=w.lane_index) for w in p] for p in o.waypoint_paths]
        ev = REvents(**o.events._asdict())
        return RObservation(dt=o.dt, step_count=o.step_count, elapsed_sim_time=o.elapsed_sim_time, events=ev,
                            ego_vehicle_state=ego, neighborhood_vehicle_states=nbs, waypoint_paths=paths,
                            distance_travelled=o.distance_travelled, lidar_point_cloud=None,
                            drivable_area_grid_map=None, occupancy_grid_map=None, top_down_rgb=None,
                            road_waypoints=None, via_data=RVias(near_via_points=[], hit_via_points=[]))

    rng = np.random.default_rng(7)
    out = {}
    rows = ob.reset_observe()
    tick = 0
    for t in range(12):
        acts = np.where(rng.random((E, N)) < 0.6, 0, rng.integers(1, 4, (E, N))).astype(np.int8)
        rows = ob.step(acts)
        if t % 4 != 3:
            continue
        for k, v in rows.items():
            out[f"t{tick}_in_{k}"] = v
        for i in range(N):
            if not rows["active"][i]:
                continue
            ro = to_ref(builder.build(rows, i, t + 2, round((t + 2) * 0.1, 6)))
            ttc = rco.lane_ttc(ro)
            for k, v in ttc.items():
                out[f"t{tick}_a{i}_lanettc_{k}"] = np.asarray(v, dtype=np.float64)
            wp = rfo._std_waypoints(ro.waypoint_paths)
            for k, v in wp.items():
                out[f"t{tick}_a{i}_wp_{k}"] = v
            nb = rfo._std_neighbors(ro.neighborhood_vehicle_states)
            for k, v in (nb or {}).items():
                out[f"t{tick}_a{i}_nb_{k}"] = v
            sttc = rfo._std_ttc(ro)
            for k, v in (sttc or {}).items():
                out[f"t{tick}_a{i}_ttc_{k}"] = np.asarray(v)
            for k, v in rfo._std_ego(ro.ego_vehicle_state).items():
                out[f"t{tick}_a{i}_ego_{k}"] = np.asarray(v)
        tick += 1
    out["n_ticks"] = np.array(tick)
    return out


def dump_sensors(nets):
    """Reference sensor classes driven directly: AccelerometerSensor
    (sensors.py:1046-1087), DrivenPathSensor + Sensors._vehicle_is_not_moving (:830-877, :511-525),
    TripMeterSensor on the reference's own road network (:880-947) and
    Sensors._vehicle_is_wrong_way (:581-586) over Lane.center_pose_at_point."""
    from unittest.mock import Mock

    from smarts.core.coordinates import Heading, Point, Pose
    from smarts.core.sensors import AccelerometerSensor, DrivenPathSensor, Sensors, TripMeterSensor

    out = {}
    rng = np.random.default_rng(11)
    # ---- accelerometer
    acc = AccelerometerSensor(None)
    lv, av = rng.normal(size=(8, 3)), rng.normal(size=(8, 3))
    out["acc_lv"], out["acc_av"] = lv, av
    out["acc_out"] = np.array([np.concatenate(acc(lv[i], av[i], 0.1)) for i in range(8)])
    # ---- driven path / not moving: 3 s of motion, then creeping, window 2 s, threshold 1 m
    veh = Mock()
    sim = Mock()
    dp = DrivenPathSensor(veh, max_path_length=500)
    veh.driven_path_sensor = dp
    pos = np.zeros(2)
    xs, ys, ts, dist, flag = [], [], [], [], []
    for t in range(80):
        step = 0.9 if t < 30 else (0.04 if t < 60 else 0.06)
        pos = pos + step * np.array([math.cos(0.02 * t), math.sin(0.02 * t)])
        veh.position = np.array([pos[0], pos[1], 0.0])
        sim.elapsed_sim_time = round((t + 2) * 0.1, 3)
        dp.track_latest_driven_path(sim)
        xs.append(pos[0]); ys.append(pos[1]); ts.append(sim.elapsed_sim_time)
        dist.append(float(dp.distance_travelled(sim, last_n_seconds=2.0)))
        flag.append(bool(Sensors._vehicle_is_not_moving(sim, veh, 2.0, 1.0)))
    out["dp_x"], out["dp_y"], out["dp_t"] = np.array(xs), np.array(ys), np.array(ts)
    out["dp_dist"], out["dp_not_moving"] = np.array(dist), np.array(flag)
    # ---- trip meter on scenarios/loop: forward along a lane, a sideways hop, a reverse hop
    rn = make_reference_road_network(nets["loop"])
    lane = rn.lane_by_id(sorted(l.getID() for l in nets["loop"].all_lanes() if not l.getID().startswith(":"))[0])
    poses = []
    for k in range(24):
        off = 5.0 + 1.3 * k
        if k == 17:
            off -= 4.0  # one step backwards
        p = lane.from_lane_coord(__import__("smarts.core.coordinates", fromlist=["RefLinePoint"]).RefLinePoint(s=off, t=0.9 if k in (9, 10) else 0.0))
        vec = lane.vector_at_offset(off)
        h = math.atan2(vec[1], vec[0]) - math.pi / 2
        poses.append((float(p.x), float(p.y), float((h + math.pi) % (2 * math.pi) - math.pi)))
    vehicle = Mock()
    vehicle.length = 3.68
    plan = Mock()
    plan.mission.has_fixed_route = False
    tsim = Mock()
    tsim.road_map = rn
    x0, y0, h0 = poses[0]
    vehicle.pose = Pose.from_center((x0, y0, 0), Heading(h0))
    tm = TripMeterSensor(vehicle, tsim, plan)
    first = tm._wps_for_distance[0] if tm._wps_for_distance else None
    out["trip_poses"] = np.array(poses)
    out["trip_first_wp"] = np.array([first.pos[0], first.pos[1], float(first.heading)]) if first else np.zeros(0)
    totals, incs = [], []
    for (x, y, h) in poses:
        wps = rn.waypoint_paths(Pose.from_center((x, y, 0), Heading(h)), lookahead=32, route=None)
        tm.append_waypoint_if_new(wps[0][0])
        totals.append(float(tm()))
        incs.append(float(tm(increment=True)))
    out["trip_total"], out["trip_inc"] = np.array(totals), np.array(incs)
    # ---- wrong way
    for name, net in nets.items():
        rn = make_reference_road_network(net) if name != "loop" else rn
        ps = sample_poses(net, np.random.default_rng(5), 120, heading_noise=2.0)
        rows = []
        for (x, y, h) in ps:
            lane = rn.nearest_lane(Point(x, y, 0), radius=7.0)
            if lane is None or lane.in_junction:
                continue
            v = Mock()
            v.pose = Pose.from_center((x, y, 0), Heading(h))
            target = lane.center_pose_at_point(Point(x, y, 0)).heading
            rows.append((x, y, h, float(target), float(bool(Sensors._vehicle_is_wrong_way(v, lane)))))
        out[f"ww_{name}"] = np.array(rows)
        out[f"ww_{name}_lanes"] = np.array([rn.nearest_lane(Point(r[0], r[1], 0), radius=7.0).lane_id for r in rows])
    return out


def dump_trajectory_pd(rn, net, rng, n):
    """Reference TrajectoryTrackingController.perform_trajectory_tracking_PD
    (trajectory_tracking_controller.py:230-331) on mock vehicles; trajectories are the reference's
    own waypoint paths (positions, headings) with a speed profile, as a Tracker agent would send."""
    import yaml

    from smarts.core.chassis import AckermannChassis
    from smarts.core.controllers.trajectory_tracking_controller import (
        TrajectoryTrackingController,
        TrajectoryTrackingControllerState,
    )
    from smarts.core.coordinates import Heading, Pose
    from smarts.core.utils.math import fast_quaternion_from_angle

    with open(os.path.join(REF, "smarts", "core", "models", "controller_parameters.yaml")) as f:
        params = yaml.safe_load(f)["sedan"]["control"]
    poses = (sample_poses(net, rng, n // 2, lateral=1.0, heading_noise=0.2, far_fraction=0.0)
             + sample_poses(net, rng, n - n // 2, lateral=0.2, heading_noise=0.03, far_fraction=0.0))
    TMAX = 11
    cols = {k: [] for k in ["x", "y", "heading", "speed", "lat_speed", "yaw_z", "n", "traj",
                            "in_state", "throttle", "brake", "steering", "out_state"]}
    for (x, y, h) in poses:
        hd = Heading(h)
        pose = Pose(position=np.array([x, y, 0.01265]), orientation=fast_quaternion_from_angle(hd), heading_=hd)
        paths = rn.waypoint_paths(pose, lookahead=int(rng.choice([3, 8, 12, 32])), route=None)
        if not paths:
            continue
        path = paths[int(rng.integers(len(paths)))]
        npts = len(path)
        v0 = float(rng.uniform(0.0, 25.0))
        speeds = [max(0.0, v0 + float(rng.normal(0, 0.5)) * i / max(npts - 1, 1)) for i in range(npts)]
        trajectory = [[float(w.pos[0]) for w in path], [float(w.pos[1]) for w in path],
                      [float(w.heading) for w in path], speeds]
        speed = float(rng.uniform(0.0, 24.0)) if rng.random() < 0.9 else 0.0
        lat = float(rng.normal(0, 0.3))
        yawz = float(rng.normal(0, 0.2))
        chassis = AckermannChassis.__new__(AckermannChassis)
        chassis.__dict__["longitudinal_lateral_speed"] = (math.sqrt(max(speed * speed - lat * lat, 0.0)), lat)
        chassis.__dict__["velocity_vectors"] = (np.array([0.0, 0.0, 0.0]), np.array([0.0, 0.0, yawz]))
        chassis.__dict__["_controller_parameters"] = params
        captured = {}

        def control(throttle=0, brake=0, steering=0, captured=captured):
            captured.update(throttle=float(throttle), brake=float(brake), steering=float(steering))

        vehicle = types.SimpleNamespace(chassis=chassis, pose=pose, position=pose.position, heading=hd, speed=speed,
                                        control=control)
        st = TrajectoryTrackingControllerState()
        if rng.random() < 0.8:
            st.heading_error = float(rng.normal(0, 0.1))
            st.lateral_error = float(rng.normal(0, 0.3))
            st.velocity_error = float(rng.normal(0, 1.0))
            st.integral_velocity_error = float(rng.normal(0, 2.0))
            st.integral_windup_error = float(rng.normal(0, 0.2))
            st.steering_state = float(np.clip(rng.normal(0, 0.3), -1, 1))
            st.throttle_state = float(rng.uniform(-1, 1))
        fields = ("heading_error", "lateral_error", "velocity_error", "integral_velocity_error", "integral_windup_error",
                  "steering_state", "throttle_state")
        in_state = [float(getattr(st, k)) for k in fields]
        TrajectoryTrackingController.perform_trajectory_tracking_PD(trajectory, vehicle, st, 0.1)
        # what travels to the device: the first ten points and the last one, plus the true length
        packed = np.zeros((4, TMAX))
        for r in range(4):
            head = trajectory[r][:10]
            packed[r, :len(head)] = head
            packed[r, 10] = trajectory[r][-1]
        for k, v in dict(x=x, y=y, heading=h, speed=speed, lat_speed=lat, yaw_z=yawz, n=npts, traj=packed,
                         in_state=in_state, out_state=[float(getattr(st, k)) for k in fields],
                         throttle=captured["throttle"], brake=captured["brake"], steering=captured["steering"]).items():
            cols[k].append(v)
    return {k: np.array(v) for k, v in cols.items()}


def dump_via_sensor(net):
    """Reference ViaSensor (sensors.py:1090-1149) on scenarios/intersections/4lane with the via points of
    that scenario's EndlessMission (scenario.py:24-73), resolved as Scenario.to_scenario_via does
    (scenario.py:652-676), for a vehicle driven through them at various speeds."""
    from unittest.mock import Mock

    from smarts.core.coordinates import RefLinePoint
    from smarts.core.plan import Via as PlanVia
    from smarts.core.sensors import ViaSensor

    rn = make_reference_road_network(net)
    spec = [("edge-south-SN", 1, 30, 4), ("edge-west-EW", 0, 20, 8), ("edge-west-EW", 1, 50, 2), ("edge-west-EW", 0, 55, 5),
            ("edge-west-EW", 1, 60, 2), ("edge-west-EW", 0, 65, 2), ("edge-west-EW", 1, 70, 2)]
    vias = []
    for road_id, lane_index, off, speed in spec:
        lane = rn.road_by_id(road_id).lane_at_index(lane_index)
        pos = lane.from_lane_coord(RefLinePoint(off))
        vias.append(PlanVia(lane_id=lane.lane_id, road_id=road_id, lane_index=lane_index, position=tuple(pos[:2]),
                            hit_distance=lane.width_at_offset(off) / 2, required_speed=speed))
    plan = Mock()
    plan.mission.via = tuple(vias)
    plan.road_map = rn
    vehicle = Mock()
    sensor = ViaSensor(vehicle, plan, lane_acquisition_range=40, speed_accuracy=1.5)
    # drive up edge-south-SN lane 1, then along edge-west-EW weaving between its two lanes
    track = []
    lane = rn.road_by_id("edge-south-SN").lane_at_index(1)
    for off in np.arange(5.0, 55.0, 1.7):
        p = lane.from_lane_coord(RefLinePoint(float(off)))
        track.append((float(p[0]), float(p[1]), 4.3 if off < 33 else 9.0))
    for k, off in enumerate(np.arange(2.0, 95.0, 1.3)):
        lane = rn.road_by_id("edge-west-EW").lane_at_index(int(k // 9) % 2)
        p = lane.from_lane_coord(RefLinePoint(float(off)))
        track.append((float(p[0]), float(p[1]), [8.2, 2.5, 5.9, 1.2][int(k // 12) % 4]))
    near_out, hit_out = [], []
    for (x, y, speed) in track:
        vehicle.position = np.array([x, y, 0.0])
        vehicle.speed = speed
        near, hit = sensor()
        idx = {(v.position, v.lane_index, v.required_speed): i for i, v in enumerate(vias)}
        near_out.append([idx[(p.position, p.lane_index, p.required_speed)] for p in near] + [-1] * (len(vias) - len(near)))
        hit_out.append([1 if any(h.position == v.position for h in hit) else 0 for v in vias])
    return dict(
        via_lane_ids=np.array([v.lane_id for v in vias]), via_pos=np.array([v.position for v in vias]),
        via_hit_distance=np.array([v.hit_distance for v in vias]), via_speed=np.array([float(v.required_speed) for v in vias]),
        via_spec=np.array([[s_[1], s_[2], s_[3]] for s_ in spec], dtype=np.float64), via_roads=np.array([s_[0] for s_ in spec]),
        track=np.array(track), near=np.array(near_out), hit=np.array(hit_out))


def dump_missions(rn, net, rng, name):
    """Fixed-route missions (plan.py:316-349): the reference's ``generate_routes`` (:711-765, over the
    restated ``getShortestPath`` of smarts_amd.sumo_map), ``waypoint_paths(pose, lookahead, route)``
    (:815-882), ``Sensors._vehicle_is_off_route_and_wrong_way`` (sensors.py:527-578) on mock
    sim / vehicle objects, ``TripMeterSensor`` with a fixed route (:880-944) and
    ``PositionalGoal.is_reached`` (plan.py:116-120)."""
    from unittest.mock import Mock

    from smarts.core.coordinates import Dimensions, Heading, Point, Pose
    from smarts.core.plan import PositionalGoal
    from smarts.core.sensors import Sensors, TripMeterSensor
    from smarts.core.utils.math import fast_quaternion_from_angle

    lane_ids = sorted(l.getID() for l in net.all_lanes())
    lane_no = {lid: i for i, lid in enumerate(lane_ids)}
    normal = [e for e in net.getEdges(False)]
    # routes: random reachable (start, end) pairs, plus a single-road route
    routes, pairs = [], []
    tries = 0
    want = {"loop": 3, "4lane": 8, "minicity": 8}[name]
    while len(routes) < want and tries < 400:
        tries += 1
        a, b = (normal[i] for i in rng.integers(len(normal), size=2))
        if len(routes) == 0:
            b = a
        r = rn.generate_routes(rn.road_by_id(a.getID()), rn.road_by_id(b.getID()))[0]
        ids = [road.road_id for road in r.roads]
        if not ids or (name == "minicity" and len(ids) > 60) or ids in [x[0] for x in routes]:
            continue
        routes.append((ids, r))
        pairs.append((a.getID(), b.getID()))
    out = dict(lane_ids=np.array(lane_ids), n_routes=np.array(len(routes)),
               route_off=np.cumsum([0] + [len(ids) for ids, _ in routes]).astype(np.int32),
               route_roads=np.array([rid for ids, _ in routes for rid in ids]),
               route_pairs=np.array(pairs))

    def make_pose(x, y, h):
        return Pose(position=np.array([x, y, 0.0]), orientation=fast_quaternion_from_angle(Heading(h)), heading_=Heading(h))

    class _LaneSet:
        def __init__(self, lanes):
            self._lanes = lanes

        def all_lanes(self):
            return self._lanes

    all_lanes = net.all_lanes()
    poses, pose_route = [], []
    for k, (ids, r) in enumerate(routes):
        on_route = [l for l in all_lanes if l.getEdge().getID() in ids]
        # the oncoming twins of the route's roads ("-id" <-> "id", "...-NS" <-> "...-SN") and their neighbours
        def twin(eid):
            if eid.startswith("-"):
                return eid[1:]
            for a_, b_ in (("NS", "SN"), ("SN", "NS"), ("EW", "WE"), ("WE", "EW")):
                if eid.endswith(a_):
                    return eid[: -2] + b_
            return "-" + eid
        twins = {twin(e) for e in ids}
        oncoming = [l for l in all_lanes if l.getEdge().getID() in twins]
        n_on, n_tw, n_any = (40, 25, 15) if name != "minicity" else (30, 20, 10)
        part = sample_poses(_LaneSet(on_route), rng, n_on, lateral=2.5, heading_noise=0.5, far_fraction=0.05)
        if oncoming:
            part += sample_poses(_LaneSet(oncoming), rng, n_tw, lateral=3.0, heading_noise=0.5, far_fraction=0.0)
        part += sample_poses(net, rng, n_any)
        poses += part
        pose_route += [k] * len(part)
    out["poses"] = np.array(poses)
    out["pose_route"] = np.array(pose_route, dtype=np.int32)

    # ---- waypoint paths along the route
    for lookahead in (16, 32):
        rec = dict(path_off=[0], wp_off=[0], x=[], y=[], heading=[], lane=[], lane_index=[], width=[], speed=[])
        for (x, y, h), k in zip(poses, pose_route):
            paths = rn.waypoint_paths(make_pose(x, y, h), lookahead=lookahead, route=routes[k][1])
            for p_ in paths:
                for wp in p_:
                    rec["x"].append(float(wp.pos[0]))
                    rec["y"].append(float(wp.pos[1]))
                    rec["heading"].append(float(wp.heading))
                    rec["lane"].append(lane_no[wp.lane_id])
                    rec["lane_index"].append(int(wp.lane_index))
                    rec["width"].append(float(wp.lane_width))
                    rec["speed"].append(float(wp.speed_limit))
                rec["wp_off"].append(len(rec["x"]))
            rec["path_off"].append(len(rec["wp_off"]) - 1)
        for key, v in rec.items():
            out[f"wp{lookahead}_{key}"] = np.array(v)

    # ---- off-route / wrong-way
    off, wrong = [], []
    for (x, y, h), k in zip(poses, pose_route):
        pose = make_pose(x, y, h)
        vehicle = Mock()
        vehicle.id = "v"
        vehicle.position = pose.position
        vehicle.pose = pose
        vehicle.chassis.dimensions = Dimensions(length=3.68, width=1.47, height=1.4)
        sim = Mock()
        sim.scenario.road_map = rn
        sim.vehicle_index.sensor_state_for_vehicle_id.return_value.plan.route = routes[k][1]
        o_, w_ = Sensors._vehicle_is_off_route_and_wrong_way(sim, vehicle)
        off.append(bool(o_))
        wrong.append(bool(w_))
    out["off_route"] = np.array(off, dtype=np.uint8)
    out["wrong_way"] = np.array(wrong, dtype=np.uint8)

    # ---- trip meter with a fixed route: a drive along the route that leaves it sideways and returns
    ids, r = routes[-1]
    track = []
    for rid in ids[:6]:
        for lane in net.getEdge(rid).getLanes()[:1]:
            shape = np.asarray(lane.getShape(False), dtype=np.float64)
            total = float(np.sqrt(((shape[1:] - shape[:-1]) ** 2).sum(axis=1)).sum())
            from smarts_amd.sumo_map import polyline_point_at
            for s_ in np.arange(0.3, total, 1.9):
                x, y = polyline_point_at(shape, float(s_))
                x2, y2 = polyline_point_at(shape, min(float(s_) + 0.25, total))
                h = math.atan2(y2 - y, x2 - x) - math.pi / 2 if (x2, y2) != (x, y) else 0.0
                track.append((float(x), float(y), float(h)))
    # three points of every twelve are replaced by points on lanes of roads that are not on the route
    off_lanes = [l for l in all_lanes if l.getEdge().getID() not in ids]
    strays = sample_poses(_LaneSet(off_lanes), rng, len(track), lateral=0.5, heading_noise=0.1, far_fraction=0.0)
    track = [(strays[i] if (i % 12) >= 9 else p_) for i, p_ in enumerate(track)][:160]
    vehicle = Mock()
    vehicle.pose = make_pose(*track[0])
    vehicle.length = 3.68
    sim = Mock()
    sim.road_map = rn
    plan = Mock()
    plan.mission.has_fixed_route = True
    plan.route = r
    meter = TripMeterSensor(vehicle, sim, plan)
    dist, incr, counted = [], [], []
    for (x, y, h) in track:
        # the query of an agent without the waypoints sensor (sensors.py:271-275): not bound to the route
        paths = rn.waypoint_paths(make_pose(x, y, h), lookahead=1, within_radius=3.68)
        if paths:
            meter.append_waypoint_if_new(paths[0][0])
        dist.append(float(meter()))
        incr.append(float(meter(increment=True)))
        counted.append(len(meter._wps_for_distance))
    out["trip_track"] = np.array(track)
    out["trip_route"] = np.array(len(routes) - 1)
    out["trip_dist"] = np.array(dist)
    out["trip_incr"] = np.array(incr)
    out["trip_counted"] = np.array(counted, dtype=np.int32)

    # ---- PositionalGoal.is_reached
    goal = PositionalGoal(position=Point(float(poses[0][0]), float(poses[0][1]), 0.0), radius=2.0)
    probes = []
    for ang in np.linspace(0, 2 * math.pi, 13):
        for rad in (0.0, 1.9999, 2.0, 2.0001, 3.5):
            probes.append((poses[0][0] + rad * math.cos(ang), poses[0][1] + rad * math.sin(ang)))
    reached = []
    for (x, y) in probes:
        v = Mock()
        v.position = np.array([x, y, 0.0])
        reached.append(bool(goal.is_reached(v)))
    out["goal"] = np.array([poses[0][0], poses[0][1], 2.0])
    out["goal_probes"] = np.array(probes)
    out["goal_reached"] = np.array(reached, dtype=np.uint8)
    return out


def dump_road_waypoints(rn, net, rng, name):
    """The reference's RoadWaypointsSensor (sensors.py:991-1040) on mock vehicle / sim / plan objects, with the
    endless mission's empty route and with a fixed route.  Poses whose nearest lane is junction-internal make the
    reference raise (Road.parallel_roads asks sumolib for the internal edge's from-node, which is None): they
    are recorded as such."""
    from unittest.mock import Mock

    from smarts.core.coordinates import Heading, Pose
    from smarts.core.sensors import RoadWaypointsSensor
    from smarts.core.utils.math import fast_quaternion_from_angle

    lane_ids = sorted(l.getID() for l in net.all_lanes())
    lane_no = {lid: i for i, lid in enumerate(lane_ids)}
    n = {"loop": 40, "4lane": 60, "minicity": 60}[name]
    poses = sample_poses(net, rng, n, lateral=1.5, heading_noise=0.3, far_fraction=0.05)
    # a fixed route for the second half of the poses: through the first reachable pair of roads
    normal = net.getEdges(False)
    route = None
    for _ in range(200):
        a, b = (normal[i] for i in rng.integers(len(normal), size=2))
        r = rn.generate_routes(rn.road_by_id(a.getID()), rn.road_by_id(b.getID()))[0]
        if len(r.roads) >= 3:
            route = r
            break
    route_ids = [road.road_id for road in route.roads]
    if name != "loop":
        on_route = [l for l in net.all_lanes() if l.getEdge().getID() in route_ids and not l.getEdge().isSpecial()]

        class _LaneSet:
            def all_lanes(self):
                return on_route

        poses[n // 2:] = sample_poses(_LaneSet(), rng, n - n // 2, lateral=1.5, heading_noise=0.3, far_fraction=0.0)
    rec = dict(lane_off=[0], lane=[], path_off=[0], wp_off=[0], x=[], y=[], heading=[], wp_lane=[], lane_index=[], width=[], speed=[])
    raised, routed = [], []
    for i, (x, y, h) in enumerate(poses):
        hd = Heading(h)
        pose = Pose(position=np.array([x, y, 0.0]), orientation=fast_quaternion_from_angle(hd), heading_=hd)
        vehicle = Mock()
        vehicle.pose = pose
        vehicle.position = pose.position
        vehicle.heading = hd
        sim = Mock()
        sim.road_map = rn
        plan = Mock()
        use_route = i >= n // 2
        plan.route = route if use_route else rn.empty_route()
        routed.append(1 if use_route else 0)
        sensor = RoadWaypointsSensor(vehicle, sim, plan, horizon=32)
        try:
            lanes = sensor().lanes
            raised.append(0)
        except AttributeError:
            lanes = {}
            raised.append(1)
        for lane_id, paths in lanes.items():
            rec["lane"].append(lane_no[lane_id])
            for p_ in paths:
                for wp in p_:
                    rec["x"].append(float(wp.pos[0]))
                    rec["y"].append(float(wp.pos[1]))
                    rec["heading"].append(float(wp.heading))
                    rec["wp_lane"].append(lane_no[wp.lane_id])
                    rec["lane_index"].append(int(wp.lane_index))
                    rec["width"].append(float(wp.lane_width))
                    rec["speed"].append(float(wp.speed_limit))
                rec["wp_off"].append(len(rec["x"]))
            rec["path_off"].append(len(rec["wp_off"]) - 1)
        rec["lane_off"].append(len(rec["lane"]))
    out = {k: np.array(v) for k, v in rec.items()}
    out.update(lane_ids=np.array(lane_ids), poses=np.array(poses), raised=np.array(raised, dtype=np.uint8),
               routed=np.array(routed, dtype=np.uint8), route_roads=np.array(route_ids), horizon=np.array(32))
    return out


def dump_default_missions(nets):
    """What hiway-v0 assigns agents of a scenario without missions.pkl: the reference's own
    Mission.random_endless_mission (plan.py:225-249) over SumoRoadNetwork.random_route (sumo_road_network.py:803-810),
    four in a row from CPython's `random` stream after smarts.core.seed(42) — straight after the seeding (`rolls0`) and
    after the three random.randint rolls Scenario.scenario_variations draws for one shuffled scenario root
    (scenario.py:211-214; `rolls3`: the stream position at which hiway-v0's TrapManager asks)."""
    import random

    import smarts.core
    from smarts.core.plan import Mission

    out = {}
    for name, net in nets.items():
        rn = make_reference_road_network(net)
        for rolls in (0, 3):
            smarts.core.seed(42)
            for _ in range(rolls):
                random.randint(0, 1)
            ms = [Mission.random_endless_mission(rn) for _ in range(4)]
            out[f"{name}_rolls{rolls}_position"] = np.array([np.asarray(m.start.position, dtype=np.float64)[:2] for m in ms])
            out[f"{name}_rolls{rolls}_heading"] = np.array([float(m.start.heading) for m in ms])
            assert all(m.goal.is_endless() for m in ms)
    return out


def main():
    install_reference()
    from smarts_amd.sumo_map import load_net

    logging.basicConfig(level=logging.WARNING)
    if os.environ.get("GOLDEN_ONLY", "") in ("", "lidar"):
        np.savez_compressed(os.path.join(OUT, "lidar_rays.npz"), **dump_lidar_rays())
        print("lidar rays written")
    if os.environ.get("GOLDEN_ONLY", "") in ("", "stdobs"):
        np.savez_compressed(os.path.join(OUT, "std_obs.npz"), **dump_std_obs())
        print("std obs written")
    if os.environ.get("GOLDEN_ONLY", "") in ("", "sensors"):
        nets = {n: load_net(os.path.join(REF, rel)) for n, rel in SCENARIOS.items()}
        np.savez_compressed(os.path.join(OUT, "sensors.npz"), **dump_sensors(nets))
        print("sensor goldens written")
    if os.environ.get("GOLDEN_ONLY", "") in ("", "trajectory"):
        net = load_net(os.path.join(REF, SCENARIOS["minicity"]))
        np.savez_compressed(os.path.join(OUT, "trajectory_pd.npz"),
                            **dump_trajectory_pd(make_reference_road_network(net), net, np.random.default_rng(21), 260))
        print("trajectory PD goldens written")
    if os.environ.get("GOLDEN_ONLY", "") in ("", "vias"):
        np.savez_compressed(os.path.join(OUT, "via_sensor.npz"), **dump_via_sensor(load_net(os.path.join(REF, SCENARIOS["4lane"]))))
        print("via sensor goldens written")
    if os.environ.get("GOLDEN_ONLY", "") in ("", "missions"):
        for name, rel in SCENARIOS.items():
            net = load_net(os.path.join(REF, rel))
            ms = dump_missions(make_reference_road_network(net), net, np.random.default_rng(4100 + len(name)), name)
            np.savez_compressed(os.path.join(OUT, f"missions_{name}.npz"), **ms)
            print(name, "missions: routes", int(ms["n_routes"]), "poses", len(ms["poses"]), "off-route",
                  int(ms["off_route"].sum()), "wrong-way", int(ms["wrong_way"].sum()))
    if os.environ.get("GOLDEN_ONLY", "") in ("", "default_missions"):
        nets = {n: load_net(os.path.join(REF, rel)) for n, rel in SCENARIOS.items()}
        np.savez_compressed(os.path.join(OUT, "default_missions.npz"), **dump_default_missions(nets))
        print("default missions written")
    if os.environ.get("GOLDEN_ONLY", "") in ("", "roadwp"):
        for name, rel in SCENARIOS.items():
            net = load_net(os.path.join(REF, rel))
            rw = dump_road_waypoints(make_reference_road_network(net), net, np.random.default_rng(5200 + len(name)), name)
            np.savez_compressed(os.path.join(OUT, f"road_waypoints_{name}.npz"), **rw)
            print(name, "road waypoints: poses", len(rw["poses"]), "lanes", len(rw["lane"]), "paths", len(rw["wp_off"]) - 1,
                  "raised", int(rw["raised"].sum()))
    if os.environ.get("GOLDEN_ONLY", "") in ("lidar", "stdobs", "sensors", "trajectory", "vias", "missions", "roadwp", "default_missions"):
        return
    for name, rel in SCENARIOS.items():
        net = load_net(os.path.join(REF, rel))
        rn = make_reference_road_network(net)
        lp = dump_lanepoints(rn)
        print(name, "lanepoints:", len(lp["x"]))
        if name == "minicity":
            # the full table is large; keep a digest + a slice
            keep = 4000
            digest = {k: (float(np.sum(v.astype(np.float64))) if v.dtype.kind in "fiu" else None) for k, v in lp.items()}
            lp_small = {k: (v[:keep] if k not in ("lane_ids", "next_off", "next_idx") else v) for k, v in lp.items()}
            lp_small["next_off"] = lp["next_off"][: keep + 1]
            lp_small["next_idx"] = lp["next_idx"][: lp["next_off"][keep]]
            lp_small["total"] = np.array(len(lp["x"]))
            lp_small["sum_x"] = np.array(digest["x"])
            lp_small["sum_y"] = np.array(digest["y"])
            lp_small["sum_heading"] = np.array(digest["heading"])
            np.savez_compressed(os.path.join(OUT, f"lanepoints_{name}.npz"), **lp_small)
        else:
            np.savez_compressed(os.path.join(OUT, f"lanepoints_{name}.npz"), **lp)
        rng = np.random.default_rng(20240 + len(name))
        n = 300 if name != "minicity" else 200
        poses = sample_poses(net, rng, n)
        for lookahead in (16, 32):
            for route_kind in ("empty_route", "none"):
                if route_kind == "none" and lookahead == 16:
                    continue
                wp = dump_waypoint_paths(rn, poses, lookahead, route_kind)
                np.savez_compressed(os.path.join(OUT, f"waypoints_{name}_{route_kind}_{lookahead}.npz"), **wp)
                print(name, route_kind, lookahead, "paths:", len(wp["wp_off"]) - 1)
        nr = dump_nearest(rn, poses)
        np.savez_compressed(os.path.join(OUT, f"nearest_{name}.npz"), **nr)
        ct = dump_controller(rn, net, np.random.default_rng(777 + len(name)), 200 if name != "minicity" else 80)
        np.savez_compressed(os.path.join(OUT, f"controller_{name}.npz"), **ct)
        print(name, "controller cases:", len(ct["x"]))


if __name__ == "__main__":
    main()
