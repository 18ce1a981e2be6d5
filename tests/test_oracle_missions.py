"""Fixed-route missions, oracle against the reference-generated fixtures ``tests/golden/missions_<map>.npz``
(tests/golden/gen_golden.py dump_missions: the reference's generate_routes, waypoint_paths with a route,
_vehicle_is_off_route_and_wrong_way, TripMeterSensor with a fixed route, PositionalGoal.is_reached)."""
import os
import types

import numpy as np
import pytest

from conftest import GOLDEN
from oracle.road_network import OLanePoints
from test_oracle_golden import _paths_equal

MAP_NAMES = ["loop", "4lane", "minicity"]


def _golden(name):
    return np.load(os.path.join(GOLDEN, f"missions_{name}.npz"))


def _routes(g):
    off = g["route_off"]
    return [[str(r) for r in g["route_roads"][off[k]:off[k + 1]]] for k in range(int(g["n_routes"]))]


def test_route_known_answer_of_the_reference_tests(oracle_maps):
    """test_map.py:123-125: the route edge-north-NS -> edge-east-WE of the 4lane map has 4 roads."""
    om = oracle_maps("4lane")
    roads = om.generate_routes(om.road_by_id("edge-north-NS"), om.road_by_id("edge-east-WE"))
    assert len(roads) == 4 and roads[0] == "edge-north-NS" and roads[-1] == "edge-east-WE"
    assert all(r.startswith(":junction-intersection") for r in roads[1:3])


@pytest.mark.parametrize("name", MAP_NAMES)
def test_generate_routes_matches_reference(name, oracle_maps):
    g, om = _golden(name), oracle_maps(name)
    routes = _routes(g)
    assert len(routes) >= 3
    for (a, b), want in zip(g["route_pairs"], routes):
        assert om.generate_routes(om.road_by_id(str(a)), om.road_by_id(str(b))) == want, (a, b)
    assert len(routes[0]) == 1  # the single-road route


@pytest.mark.parametrize("name", MAP_NAMES)
@pytest.mark.parametrize("lookahead", [16, 32])
def test_waypoint_paths_along_route_match_reference(name, lookahead, oracle_maps):
    import tie_sensitive

    g, om = _golden(name), oracle_maps(name)
    routes = _routes(g)
    w = {k[len(f"wp{lookahead}_"):]: g[k] for k in g.files if k.startswith(f"wp{lookahead}_")}
    lane_ids = [str(x) for x in g["lane_ids"]]
    tie_sensitive_poses = []
    for i, ((px, py, ph), k) in enumerate(zip(g["poses"], g["pose_route"])):
        pos = np.array([px, py, 0.0])
        OLanePoints.tie_rule = "kdtree"
        try:
            ok, err = _paths_equal(om.waypoint_paths(pos, ph, lookahead, route=routes[k]), w, i, lane_ids)
        finally:
            OLanePoints.tie_rule = "index"
        assert ok and err == 0.0, f"pose {i}: oracle (kdtree ties) != reference"
        ok2, err2 = _paths_equal(om.waypoint_paths(pos, ph, lookahead, route=routes[k]), w, i, lane_ids)
        if not (ok2 and err2 == 0.0):
            tie_sensitive_poses.append(i)
    assert tie_sensitive_poses == tie_sensitive.ROUTE_WAYPOINTS[(name, lookahead)], tie_sensitive_poses


@pytest.mark.parametrize("name", MAP_NAMES)
def test_off_route_and_wrong_way_match_reference(name, oracle_maps):
    from oracle.sim import OracleEnv

    g, om = _golden(name), oracle_maps(name)
    routes = _routes(g)
    env = types.SimpleNamespace(road_map=om)
    assert 0 < g["off_route"].sum() < len(g["poses"]) and g["wrong_way"].sum() > 0
    for i, ((px, py, ph), k) in enumerate(zip(g["poses"], g["pose_route"])):
        b = types.SimpleNamespace(position=np.array([px, py, 0.0]), heading=ph, length=3.68, width=1.47)
        off, wrong = OracleEnv._off_route_and_wrong_way(env, b, tuple(routes[k]))
        assert (bool(off), bool(wrong)) == (bool(g["off_route"][i]), bool(g["wrong_way"][i])), i


@pytest.mark.parametrize("name", MAP_NAMES)
def test_trip_meter_with_a_fixed_route_matches_reference(name, oracle_maps):
    from oracle.sim import OracleEnv

    g, om = _golden(name), oracle_maps(name)
    route = _routes(g)[int(g["trip_route"])]
    track = g["trip_track"]
    env = types.SimpleNamespace(road_map=om)
    x, y, h = track[0]
    first = om.waypoint_paths(np.array([x, y, 0.0]), h, lookahead=1, within_radius=3.68)
    ag = types.SimpleNamespace(wps_for_distance=[first[0][0]] if first else [], dist_travelled=0.0,
                               last_dist_travelled=0.0, route=tuple(route), goal=(0.0, 0.0, 1.0))
    for k, (x, y, h) in enumerate(track):
        paths = om.waypoint_paths(np.array([x, y, 0.0]), h, lookahead=1, within_radius=3.68)
        if paths:
            OracleEnv._append_waypoint_if_new(env, ag, paths[0][0])
        assert ag.dist_travelled == g["trip_dist"][k], k
        assert ag.dist_travelled - ag.last_dist_travelled == g["trip_incr"][k], k
        assert len(ag.wps_for_distance) == g["trip_counted"][k], k
    # the drive leaves the route (waypoints not counted) and comes back
    assert g["trip_dist"][-1] > 20.0 and (np.diff(g["trip_counted"]) == 0).sum() > 3, (g["trip_dist"][-1], g["trip_counted"])


def test_positional_goal_matches_reference():
    g = _golden("loop")
    gx, gy, gr = g["goal"]
    got = [(x - gx) ** 2 + (y - gy) ** 2 <= gr ** 2 for x, y in g["goal_probes"]]
    assert np.array_equal(np.array(got, dtype=np.uint8), g["goal_reached"])
    assert 0 < g["goal_reached"].sum() < len(got)


# ---------------------------------------------------------------------------------------------
# RoadWaypointsSensor (sensors.py:991-1040): tests/golden/road_waypoints_<map>.npz
# ---------------------------------------------------------------------------------------------
def road_waypoints_equal(got, g, i, lane_ids):
    """`got` ({lane id: [paths]}) against pose i of the flattened fixture -> (equal, worst |error|)."""
    l0, l1 = g["lane_off"][i], g["lane_off"][i + 1]
    if [lane_ids.index(k) for k in got] != list(g["lane"][l0:l1]):
        return False, 0.0
    worst = 0.0
    for j, paths in enumerate(got.values()):
        p0, p1 = g["path_off"][l0 + j], g["path_off"][l0 + j + 1]
        if len(paths) != p1 - p0:
            return False, 0.0
        for k, p in enumerate(paths):
            a, b = g["wp_off"][p0 + k], g["wp_off"][p0 + k + 1]
            if len(p) != b - a or [lane_ids.index(w.lane_id) for w in p] != list(g["wp_lane"][a:b]):
                return False, 0.0
            if [w.lane_index for w in p] != list(g["lane_index"][a:b]):
                return False, 0.0
            for key, vals in (("x", [w.pos[0] for w in p]), ("y", [w.pos[1] for w in p]), ("heading", [w.heading for w in p]),
                              ("width", [w.lane_width for w in p]), ("speed", [w.speed_limit for w in p])):
                worst = max(worst, float(np.abs(np.array(vals, dtype=np.float64) - g[key][a:b]).max()))
    return True, worst


@pytest.mark.parametrize("name", MAP_NAMES)
def test_road_waypoints_match_reference(name, oracle_maps):
    import tie_sensitive
    from oracle.sensors_extra import road_waypoints

    om = oracle_maps(name)
    g = np.load(os.path.join(GOLDEN, f"road_waypoints_{name}.npz"))
    lane_ids = [str(x) for x in g["lane_ids"]]
    route = [str(r) for r in g["route_roads"]]
    answered, tie_sensitive_poses = 0, []
    for i, (x, y, h) in enumerate(g["poses"]):
        if g["raised"][i]:
            continue  # nearest lane junction-internal: the reference raises (no from-node in sumolib)
        rt = route if g["routed"][i] else None
        OLanePoints.tie_rule = "kdtree"
        try:
            ok, err = road_waypoints_equal(road_waypoints(om, (x, y, 0.0), h, 32, rt), g, i, lane_ids)
        finally:
            OLanePoints.tie_rule = "index"
        assert ok and err == 0.0, f"pose {i}: oracle (kdtree ties) != reference"
        ok2, err2 = road_waypoints_equal(road_waypoints(om, (x, y, 0.0), h, 32, rt), g, i, lane_ids)
        if not (ok2 and err2 == 0.0):
            tie_sensitive_poses.append(i)
        answered += 1
    assert answered >= 25 and g["routed"].sum() > 0
    assert tie_sensitive_poses == tie_sensitive.ROAD_WAYPOINTS[name], tie_sensitive_poses
