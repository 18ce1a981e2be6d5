"""Known-answer tests of the reference, re-expressed against the oracle and the host map loader.

Each test names the reference test it restates (values copied from there are data, not code).
"""
import math

import numpy as np

from oracle import ref_math as rm
from oracle.dynamics import VehicleBody


def test_egocentric_conversion():
    # smarts/core/utils/tests/test_math.py:27-38
    pec = rm.position_to_ego_frame([1, 2, 3], [1, -5, 2], -3)
    assert np.allclose([-0.9878400564190705, -6.929947476203118, 1.0], pec)


def test_signed_dist_to_line_doctest():
    # smarts/core/utils/math.py:168-172
    assert rm.signed_dist_to_line(np.array([2, 0]), np.array([0, 0]), np.array([0, 1.0])) == -2.0
    assert rm.signed_dist_to_line(np.array([-1.5, 0]), np.array([0, 0]), np.array([0, 1.0])) == 1.5


def test_heading_wrap():
    # smarts/core/tests/test_coordinates.py:147-176
    H = rm.wrap_heading
    rel = rm.heading_relative_to
    assert H(0) == 0
    assert H(-3.14) == -3.14
    assert H(-math.pi - 1) == math.pi - 1
    assert H(math.pi + 1) == -math.pi + 1
    assert math.isclose(rel(H(math.pi / 4), H(math.pi)), H(-2.356194490192345))
    assert math.isclose(rel(H(0), H(math.pi + 1)), H(math.pi - 1))
    assert math.isclose(rel(H(math.pi + 1), H(0)), H(-math.pi + 1))
    assert math.isclose(rel(H(math.pi + 1), H(-math.pi - 1)), H(2))
    assert math.isclose(rel(H(2 * math.pi), H(-2 * math.pi)), H(0), abs_tol=1e-12)
    assert math.isclose(rel(H(2 * math.pi), H(-2 * math.pi - 1)), H(1))
    assert math.isclose(rel(H(2 * math.pi), H(4 * math.pi)), H(0), abs_tol=1e-12)
    assert math.isclose(rel(H(-2 * math.pi), H(-4 * math.pi)), H(0), abs_tol=1e-12)


def test_front_bumper_pose():
    # smarts/core/tests/test_coordinates.py:40-72 (Pose.from_front_bumper, coordinates.py:302-321)
    origin = np.array([0.0, 1.0])
    for offset, sumo_angle, length in [([2, 0], 90, 4), ([-1, 0], 270, 2), ([5, 5], 45, math.sqrt(50) * 2),
                                        ([0, -1.5], 180, 3)]:
        front = origin + np.array(offset, dtype=float)
        heading = rm.wrap_heading((2 * math.pi - math.radians(sumo_angle)) % (2 * math.pi))  # Heading.from_sumo
        centre = front - rm.radians_to_vec(heading) * (0.5 * length)
        assert np.isclose(centre, origin, atol=2e-06).all()


def test_vehicle_bounding_box():
    # smarts/core/tests/test_vehicle.py:126-143: box 3 x 1 at (1, 1), heading 0
    b = VehicleBody(1.0, 1.0, 0.0, 0.0)
    b.length, b.width = 3.0, 1.0
    want = [[0.5, 2.5], (1.5, 2.5), (1.5, -0.5), (0.5, -0.5)]
    for got, w in zip(b.bounding_box, want):
        assert np.array_equal(got, np.array(w, dtype=float))


def test_round_param_for_dt():
    # smarts/core/utils/math.py:553-563 docstring
    assert rm.round_param_for_dt(100) == -2
    assert rm.round_param_for_dt(0.1) == 1
    assert rm.round_param_for_dt(0.01) == 2


def test_sumo_map_4lane(oracle_maps):
    # smarts/core/tests/test_map.py:48-141 on scenarios/intersections/4lane (origin-shifted as
    # `scl scenario build` does): the parts of the map API that the hot path touches.
    rmap = oracle_maps("4lane")
    point = (125.20, 139.0, 0)
    lane = rmap.nearest_lane(point)
    assert lane.lane_id == "edge-north-NS_0"
    assert lane.road.road_id == "edge-north-NS"
    assert lane.index == 0
    assert lane.length == 55.6
    s, t = lane.to_lane_coord(point)
    assert s == 1.0
    assert t == 0.0
    assert lane.width_at_offset(s) == 3.2
    assert not lane.incoming_lanes
    out_lanes = lane.outgoing_lanes
    assert len(out_lanes) == 2
    assert out_lanes[0].lane_id == ":junction-intersection_0_0"
    assert out_lanes[1].lane_id == ":junction-intersection_1_0"
    assert np.array_equal(lane.vector_at_offset(55.7), np.array([0.0, -1.0, 0.0]))
    road = rmap.road_with_point(point)
    assert road is not None and road.road_id == "edge-north-NS"
    left = [l for l in lane.road.lanes if l.index == 1][0]
    assert left.lane_id == "edge-north-NS_1"


def test_origin_shift_matches_reference_expectation(nets):
    # test_map.py:52-54 expects edge-north-NS_0 at x = 125.20 while the raw net.xml has 145.20:
    # the (-20, +130) shift of scl scenario build
    net = nets("4lane")
    assert tuple(net.shifted_by) == (-20.0, 130.0)
    assert net.getLane("edge-north-NS_0").shape[0] == (125.2, 140.0)
    assert tuple(nets("loop").shifted_by) == (0.0, 0.0)
    assert tuple(nets("minicity").shifted_by) == (-134.74, -464.69)


def test_golden_equally_spaced_vector_from_survey():
    # SURVEY.md §8c golden vector, produced by the reference on a straight lane (0,0)->(30,0):
    # point (5.3, 0.4), lookahead 8 -> first waypoints (5.3, 0), (6.2625, 0), (7.225, 0),
    # heading -pi/2, i.e. spacing (13 - 5.3) / 8.
    from oracle.road_network import LP, equally_spaced_path

    class _Lane:
        lane_id, index, _width, speed_limit = "a_0", 0, 3.2, 16.67

    lane = _Lane()
    q = rm.quat_from_angle(rm.wrap_heading(rm.vec_to_radians((1.0, 0.0))))
    pts = [LP(lane, (float(i), 0.0), q, i not in (0, 30)) for i in range(31)]
    path = pts[5:14]
    wps = equally_spaced_path(path, (5.3, 0.4), 1.0)
    assert len(wps) == 9
    assert np.allclose([w.pos[0] for w in wps[:3]], [5.3, 6.2625, 7.225])
    assert all(w.pos[1] == 0.0 for w in wps)
    assert wps[0].heading == -1.5707963267948966
    assert wps[0].lane_index == 0 and wps[0].lane_width == 3.2 and wps[0].speed_limit == 16.67


def _still(x, y, heading):
    return VehicleBody(x, y, heading, 0.0)


def test_contact_rule_reproduces_the_reference_collision_geometry():
    """smarts/core/tests/test_collision.py:86-282 restated for the footprint rule that stands in for
    Bullet's getClosestPoints(distance=0.05) (chassis.py:64-78): crossed boxes at one centre collide
    (:86-106), 10 m apart do not (:109-127), and a passenger box packed 0.0501 m clear of the ego on
    each side reports nothing (:206-282) while 0.0499 m does."""
    from oracle.dynamics import CHASSIS_LENGTH, CHASSIS_WIDTH
    from oracle.sim import COLLISION_LEEWAY, boxes_within

    assert (CHASSIS_LENGTH, CHASSIS_WIDTH) == (3.68, 1.47)  # VEHICLE_CONFIGS["passenger"] (vehicle.py:101)
    ego = _still(0.0, 0.0, -0.5 * math.pi)
    assert boxes_within(ego, _still(0.0, 0.0, 0.0), COLLISION_LEEWAY)  # a "plus": no corner inside the other box
    assert not boxes_within(ego, _still(0.0, 10.0, 0.0), COLLISION_LEEWAY)
    ego = _still(0.0, 0.0, 0.0)
    for sep, touching in ((0.0501, False), (0.0499, True)):
        ring = [(0.0, CHASSIS_LENGTH + sep), (CHASSIS_WIDTH + sep, 0.0), (0.0, -(CHASSIS_LENGTH + sep)), (-(CHASSIS_WIDTH + sep), 0.0)]
        for x, y in ring:
            assert boxes_within(ego, _still(x, y, 0.0), COLLISION_LEEWAY) is touching, (sep, x, y)
            assert boxes_within(_still(x, y, 0.0), ego, COLLISION_LEEWAY) is touching
    # rotated pair: corner of one 0.05 -/+ 1e-4 from the side of the other
    for gap, touching in ((0.0499, True), (0.0501, False)):
        h = math.pi / 4
        # the nearest corner of a box at heading pi/4 lies (L/2 + W/2) / sqrt(2) left of its centre
        reach = (0.5 * CHASSIS_LENGTH + 0.5 * CHASSIS_WIDTH) / math.sqrt(2.0)
        other = _still(0.5 * CHASSIS_WIDTH + gap + reach, 0.0, h)
        assert boxes_within(ego, other, COLLISION_LEEWAY) is touching
