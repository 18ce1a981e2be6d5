"""Host side of fixed-route missions (smarts_amd/missions.py, SumoNet.getShortestPath): CPU tests.

The planner is held to the reference-generated routes of ``tests/golden/missions_<map>.npz`` (the reference's
``generate_routes`` over the restated ``getShortestPath``), to the reference's own known answer
(``test_map.py:123-125``) and to the oracle's ``create_route``."""
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN

MAP_NAMES = ["loop", "4lane", "minicity"]


def test_shortest_path_known_answer(nets):
    """test_map.py:123-125: edge-north-NS -> edge-east-WE is a route of 4 roads (2 normal edges)."""
    from smarts_amd.missions import generate_route

    net = nets("4lane")
    edges, cost = net.getShortestPath(net.getEdge("edge-north-NS"), net.getEdge("edge-east-WE"))
    assert [e.getID() for e in edges] == ["edge-north-NS", "edge-east-WE"]
    assert cost == pytest.approx(55.6 + 55.6)
    roads = generate_route(net, "edge-north-NS", "edge-east-WE")
    assert len(roads) == 4 and roads[0] == "edge-north-NS" and roads[-1] == "edge-east-WE"
    # no way back up a one-way approach
    assert net.getShortestPath(net.getEdge("edge-east-WE"), net.getEdge("edge-north-NS"))[0] is None
    assert generate_route(net, "edge-east-WE", "edge-north-NS") == []


@pytest.mark.parametrize("name", MAP_NAMES)
def test_generate_route_matches_reference(name, nets):
    from smarts_amd.missions import generate_route

    g = np.load(os.path.join(GOLDEN, f"missions_{name}.npz"))
    off = g["route_off"]
    for k, (a, b) in enumerate(g["route_pairs"]):
        want = [str(r) for r in g["route_roads"][off[k]:off[k + 1]]]
        assert generate_route(nets(name), str(a), str(b)) == want, (a, b)


def test_plan_mission_follows_scenario_and_plan(nets, oracle_maps):
    """scenario.py:643-700 (start pose, goal radius 2) + plan.py:316-349 (route between the nearest roads
    outside junctions), against the oracle's create_route on the same points."""
    from smarts_amd.missions import Mission, Route, plan_mission

    net, om = nets("4lane"), oracle_maps("4lane")
    pm = plan_mission(net, Mission(Route(begin=("edge-north-NS", 0, 40), end=("edge-east-WE", 1, "max"))))
    lane = om.lane_by_id("edge-north-NS_0")
    x, y, _ = lane.from_lane_coord(40)
    assert pm.start_position == (x, y)
    assert pm.start_heading == pytest.approx(math.pi)  # southbound
    end = om.lane_by_id("edge-east-WE_1")
    gx, gy, _ = end.from_lane_coord(end.length - 1e-6)  # "max" (scenario.py:630-641)
    assert pm.goal == (gx, gy, 2.0)
    assert list(pm.route_roads) == om.create_route((x, y, 0.0), (gx, gy, 0.0))
    # Pose.from_front_bumper (coordinates.py:302-321): the centre lies half a length behind the start point
    cx, cy, ch = pm.spawn_pose()
    assert (cx, ch) == (pytest.approx(x), pm.start_heading) and cy == pytest.approx(y + 1.84)
    with pytest.raises(ValueError):
        plan_mission(net, Mission(Route(begin=("edge-east-WE", 0, 10), end=("edge-north-NS", 0, 10))))  # PlanningError
    with pytest.raises(ValueError):
        plan_mission(net, Mission(Route(begin=("no-such-road", 0, 10), end=("edge-north-NS", 0, 10))))


def test_load_missions_json(tmp_path):
    import json

    from smarts_amd.missions import Mission, Route, load_missions

    spec = {"a0": {"begin": ["edge-west-WE", 1, 60], "end": ["edge-east-WE", 1, "max"], "via": ["edge-west-WE"]},
            "a1": {"begin": ["edge-south-SN", 0, "base"], "end": ["edge-west-EW", 0, 20.5]}}
    p = tmp_path / "missions.json"
    p.write_text(json.dumps(spec))
    ms = load_missions(str(p))
    assert ms["a0"] == Mission(Route(("edge-west-WE", 1, 60), ("edge-east-WE", 1, "max"), ("edge-west-WE",)))
    assert ms["a1"].route.via == () and ms["a1"].route.begin == ("edge-south-SN", 0, "base")
