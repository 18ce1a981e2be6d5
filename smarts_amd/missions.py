"""Missions with fixed routes (host side): what a scenario author writes, the route the planner finds, and the
records the device reads (``include/smx.h`` ``smx_mission``).

Mirrors ``smarts/sstudio/types.py`` ``Route`` / ``Mission`` (begin / end as (road id, lane index, offset)),
``Scenario._extract_mission`` (``smarts/core/scenario.py:625-700``: start pose on the lane's centre line with
the lane's direction, ``PositionalGoal`` of radius 2 at the end), ``Plan.create_route``
(``smarts/core/plan.py:316-349``) and ``SumoRoadNetwork.generate_routes`` / ``_internal_routes_between``
(``smarts/core/sumo_road_network.py:711-800``).  The edge search underneath is ``sumolib``'s
``getShortestPath``, restated in :mod:`smarts_amd.sumo_map`.
"""
from __future__ import annotations

import json
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from .sumo_map import Edge, SumoNet
from .vias import _position_at_shape_offset

Offset = Union[float, str]
CHASSIS_LENGTH = 3.68  # vehicle.py:101 (passenger)


@dataclass(frozen=True)
class Route:
    """sstudio/types.py ``Route``: begin / end = (road id, lane index, offset in metres | "base" | "max")."""

    begin: Tuple[str, int, Offset]
    end: Tuple[str, int, Offset]
    via: Tuple[str, ...] = ()


@dataclass(frozen=True)
class Mission:
    """sstudio/types.py ``Mission`` (the route is what this path reads)."""

    route: Route


@dataclass(frozen=True)
class PlannedMission:
    """plan.py ``Mission`` + ``Plan.route`` after ``create_route``."""

    start_position: Tuple[float, float]  # Start.position: the front bumper (plan.py:39-45)
    start_heading: float
    goal: Tuple[float, float, float]     # PositionalGoal x, y, radius
    route_roads: Tuple[str, ...]         # RoadMap.Route.roads, junction-internal roads included

    def spawn_pose(self, length: float = CHASSIS_LENGTH) -> Tuple[float, float, float]:
        """Pose.from_front_bumper (coordinates.py:302-321): vehicle centre and heading."""
        # radians_to_vec (utils/math.py:262-266): heading 0 faces +y
        a = (self.start_heading + math.pi * 0.5) % (2 * math.pi)
        return (self.start_position[0] - math.cos(a) * 0.5 * length,
                self.start_position[1] - math.sin(a) * 0.5 * length, self.start_heading)


def _resolve_offset(offset: Offset, lane_length: float) -> float:
    eps = 1e-6  # scenario.py:630-641
    lane_length -= eps
    if offset == "base":
        return eps
    if offset == "max":
        return lane_length
    if offset == "random":
        raise ValueError('offset "random" is not reproducible on a fixed batch; give a number')
    return float(offset)


def _vec_to_radians(x: float, y: float) -> float:
    """utils/math.py:256-277, then ``Heading()``'s wrap into (-pi, pi] (coordinates.py:169-190)."""
    r = math.atan2(abs(y), abs(x))
    if x < 0:
        a = (r + 0.5 * math.pi) % (2 * math.pi) if y < 0 else (0.5 * math.pi - r) % (2 * math.pi)
    elif y < 0:
        a = (1.5 * math.pi - r) % (2 * math.pi)
    else:
        a = (r - 0.5 * math.pi) % (2 * math.pi)
    return a - 2 * math.pi if a > math.pi else a


def _position_and_heading(net: SumoNet, road_id: str, lane_index: int, offset: Offset):
    edge = net.getEdge(road_id)
    if edge is None:
        raise ValueError(f"unknown road {road_id!r}")
    lane = edge.getLane(lane_index)
    shape = np.asarray(lane.getShape(False), dtype=np.float64)
    length = lane.getLength()
    off = _resolve_offset(offset, length)
    x, y = _position_at_shape_offset(shape, off)
    # Lane.vector_at_offset (road_map.py:377-388)
    s_off, e_off = (length - 1, length) if off >= length else (off, off + 1)
    s_off = max(s_off, 0)
    x1, y1 = _position_at_shape_offset(shape, s_off)
    x2, y2 = _position_at_shape_offset(shape, e_off)
    return (x, y), _vec_to_radians(x2 - x1, y2 - y1)


def nearest_road_outside_junctions(net: SumoNet, point: Sequence[float]) -> Optional[Edge]:
    """``road_map.nearest_lane(point, include_junctions=False).road`` (sumo_road_network.py:676-701,
    road_map.py:91-96; the default radius max(10, 2 x 3.2))."""
    radius = max(10, 2 * 3.2)
    best, best_d = None, None
    # include_junctions=False -> getNeighboringLanes(includeJunctions=True), internal edges dropped (:682-697)
    for lane, d in net.neighboring_lanes(point[0], point[1], radius, True):
        if lane.getEdge().isSpecial():
            continue
        if best_d is None or d < best_d:  # the stable sort by distance keeps the first of equals
            best, best_d = lane, d
    return best.getEdge() if best is not None else None


def _internal_routes_between(net: SumoNet, start_edge: Edge, end_edge: Edge) -> List[List[Edge]]:
    """sumo_road_network.py:767-800."""
    routes = []
    outgoing = start_edge.getOutgoing()
    if end_edge not in outgoing:
        raise ValueError(f"{end_edge.getID()} does not follow {start_edge.getID()}")
    for connection in outgoing[end_edge]:
        conn_route = [start_edge]
        via_lane_id = connection.getViaLaneID()
        while via_lane_id:
            via_edge = net.getLane(via_lane_id).getEdge()
            conn_route.append(via_edge)
            nxt = set(c.getViaLaneID() for c in via_edge.getOutgoing()[end_edge])
            if len(nxt) != 1:
                raise ValueError(f"expected exactly one next via lane at {via_lane_id}, got {nxt}")
            via_lane_id = next(iter(nxt))
        conn_route.append(end_edge)
        routes.append(conn_route)
    return routes


def generate_route(net: SumoNet, start_road: str, end_road: str, via: Sequence[str] = ()) -> List[str]:
    """sumo_road_network.py:711-765 -> road ids of the route ([] when there is none)."""
    roads = [net.getEdge(start_road)] + [net.getEdge(v) for v in via]
    if end_road != start_road:
        roads.append(net.getEdge(end_road))
    if any(r is None for r in roads):
        raise ValueError("unknown road in route")
    edges: List[Edge] = []
    for cur, nxt in zip(roads, roads[1:] + [None]):
        if nxt is None:
            edges.append(cur)
            break
        sub = net.getShortestPath(cur, nxt)[0] or []
        if len(sub) < 2:
            return []
        edges.extend(sub[:-1])
    if len(edges) == 1:
        return [edges[0].getID()]
    used: List[str] = []
    for cur, nxt in zip(edges, edges[1:]):
        for internal_route in _internal_routes_between(net, cur, nxt):
            used.extend(e.getID() for e in internal_route)
    seen, out = set(), []
    for rid in used:  # np.unique(..., return_index=True) + sorted(indices): first occurrences in order
        if rid not in seen:
            seen.add(rid)
            out.append(rid)
    return out


def plan_mission(net: SumoNet, mission: Mission) -> PlannedMission:
    """``Scenario._extract_mission`` + ``Plan.create_route``."""
    r = mission.route
    start_pos, start_heading = _position_and_heading(net, *r.begin)
    goal_pos, _ = _position_and_heading(net, *r.end)
    start_road = nearest_road_outside_junctions(net, start_pos)
    end_road = nearest_road_outside_junctions(net, goal_pos)
    if start_road is None or end_road is None:
        raise ValueError("route must start and end in a lane")  # plan.py:330, 337
    roads = generate_route(net, start_road.getID(), end_road.getID(), r.via)
    if not roads:
        # plan.py:345-351 (PlanningError)
        raise ValueError(f"Unable to find a route between start={start_road.getID()} and end={end_road.getID()}.")
    return PlannedMission(start_pos, start_heading, (goal_pos[0], goal_pos[1], 2.0), tuple(roads))


def _polygon_offset_with_minimum_distance(point: Sequence[float], shape: np.ndarray) -> float:
    """sumolib.geomhelper.polygonOffsetWithMinimumDistanceToPoint(point, shape, perpendicular=False), whose in-tree
    twin is utils/math.py:370-390 (with line_offset_with_minimum_distance_to_point :348-367)."""
    px, py = float(point[0]), float(point[1])
    min_dist, min_offset, seen = 1e400, -1.0, 0.0
    for (x1, y1), (x2, y2) in zip(shape[:-1], shape[1:]):
        x1, y1, x2, y2 = float(x1), float(y1), float(x2), float(y2)
        ex, ey = x1 - x2, y1 - y2
        d = math.sqrt(ex * ex + ey * ey)
        u = ((px - x1) * (x2 - x1)) + ((py - y1) * (y2 - y1))
        if d == 0.0 or u < 0.0 or u > d * d:
            pos = 0.0 if u < 0.0 else d
        else:
            pos = u / d
        # position_at_offset (utils/math.py:300-316) on the segment, then the distance to it
        def close(a, b):
            return abs(a - b) <= max(1e-09 * max(abs(a), abs(b)), 0.0)
        if close(pos, 0.0):
            fx, fy = x1, y1
        elif close(d, pos):
            fx, fy = x2, y2
        else:
            fx, fy = x1 + (x2 - x1) * (pos / d), y1 + (y2 - y1) * (pos / d)
        gx, gy = px - fx, py - fy
        dist = math.sqrt(gx * gx + gy * gy)
        if dist < min_dist:
            min_dist, min_offset = dist, pos + seen
        seen += d
    return min_offset


def _offset_along_lane(shape: np.ndarray, point: Sequence[float]) -> float:
    """sumo_road_network.py:477-491."""
    px, py = float(point[0]), float(point[1])
    if not any(float(x) == px and float(y) == py for x, y in shape):
        return _polygon_offset_with_minimum_distance(point, shape)
    offset = 0.0
    for i in range(len(shape) - 1):
        if float(shape[i][0]) == px and float(shape[i][1]) == py:
            break
        ex, ey = float(shape[i][0] - shape[i + 1][0]), float(shape[i][1] - shape[i + 1][1])
        offset += math.sqrt(ex * ex + ey * ey)
    return offset


def _center_pose_at_point(lane, point: Sequence[float]):
    """Lane.center_pose_at_point (road_map.py:390-396): position on the centre line closest to `point` and the lane's
    heading there, through the quaternion as the reference's Pose holds it (coordinates.py:394-403,
    utils/math.py:78-94)."""
    shape = np.asarray(lane.getShape(False), dtype=np.float64)
    length = lane.getLength()
    off = _offset_along_lane(shape, point)
    x, y = _position_at_shape_offset(shape, off)
    s_off, e_off = (length - 1, length) if off >= length else (off, off + 1)
    s_off = max(s_off, 0)
    x1, y1 = _position_at_shape_offset(shape, s_off)
    x2, y2 = _position_at_shape_offset(shape, e_off)
    # vec_to_radians without Heading()'s wrap, then fast_quaternion_from_angle, then yaw_from_quaternion + Heading
    vx, vy = x2 - x1, y2 - y1
    r = math.atan2(abs(vy), abs(vx))
    if vx < 0:
        ang = (r + 0.5 * math.pi) % (2 * math.pi) if vy < 0 else (0.5 * math.pi - r) % (2 * math.pi)
    elif vy < 0:
        ang = (1.5 * math.pi - r) % (2 * math.pi)
    else:
        ang = (r - 0.5 * math.pi) % (2 * math.pi)
    half = ang * 0.5
    qz, qw = math.sin(half), math.cos(half)
    yaw = math.atan2(2 * (0.0 * 0.0 + qw * qz), qw * qw + 0.0 * 0.0 - 0.0 * 0.0 - qz * qz)
    heading = yaw % (2 * math.pi)
    if heading > math.pi:
        heading -= 2 * math.pi
    return (x, y), heading


def reference_spawn_table(net: SumoNet, num_envs: int, count: int, seed: int, episodes: int = 4,
                          shuffle_scenarios: bool = True) -> np.ndarray:
    """Spawn poses ``[episodes, num_envs * count, 4]`` (x, y, heading, speed 0) of agents without missions, as
    ``hiway-v0`` starts them: env ``e`` of a ParallelEnv is seeded ``seed + e`` (parallel_env.py:190-202); every
    ``reset`` draws the scenario rolls again (scenario.py:211-214), then one ``Mission.random_endless_mission`` per
    agent; the agents' vehicles, created after all missions are drawn, take a ``gen_id()`` each
    (vehicle_index.py:602: ``random.getrandbits(128)``) from the same stream.  The vehicle's centre lies half a
    chassis length behind the mission's start (Pose.from_front_bumper, vehicle.py:379-387); it stands still
    (TrapEntryTactic.default_entry_speed is None).  Episode 0 is pinned by tests/golden/default_missions.npz (the
    reference's own draw); the stream positions of later episodes follow from the calls named here (read, not run:
    hiway-v0 cannot run in this tree)."""
    import random as _random

    out = np.zeros((episodes, num_envs * count, 4), dtype=np.float64)
    for e in range(num_envs):
        rng = _random.Random(seed + e)
        for ep in range(episodes):
            ms = _draw_endless_missions(net, count, rng, 3 if shuffle_scenarios else 0)
            for i, m in enumerate(ms):
                x, y, h = m.spawn_pose()
                out[ep, e * count + i] = (x, y, h, 0.0)
            for _ in range(count):
                rng.getrandbits(128)
    return out


def random_endless_missions(net: SumoNet, count: int, seed: int, scenario_rolls: int = 3) -> List[PlannedMission]:
    """What ``hiway-v0`` gives agents of a scenario without ``missions.pkl``: ``Mission.random_endless_mission``
    (plan.py:225-249) over ``SumoRoadNetwork.random_route(1)`` (sumo_road_network.py:803-810), one per agent in
    ``agent_specs`` order (``TrapManager.init_traps``, trap_manager.py:83-92), drawn from CPython's ``random`` stream as
    ``smarts.core.seed(seed)`` leaves it (core/__init__.py:39-44).  Between the seeding and the first mission
    ``Scenario.scenario_variations`` draws three ``random.randint`` rolls for the one scenario root when scenarios
    are shuffled (scenario.py:211-214; ``HiWayEnv(shuffle_scenarios=True)`` is the default): ``scenario_rolls``.
    Returned as planned missions with an empty route (endless) and no goal."""
    import random as _random

    return _draw_endless_missions(net, count, _random.Random(seed), scenario_rolls)


def _draw_endless_missions(net: SumoNet, count: int, rng, scenario_rolls: int) -> List[PlannedMission]:
    for _ in range(scenario_rolls):
        rng.randint(0, 1)  # routes / agent missions / social agents: one candidate each in an unbuilt scenario
    out = []
    edges = net.getEdges(False)
    for _ in range(count):
        edge = rng.choice(edges)                # random_route(1).roads[0]
        lane = rng.choice(edge.getLanes())      # random.choice(road.lanes)
        offset = rng.random() * 0.3 + (0.9 - 0.3)
        offset *= lane.getLength()
        shape = np.asarray(lane.getShape(False), dtype=np.float64)
        coord = _position_at_shape_offset(shape, offset)  # n_lane.from_lane_coord(RefLinePoint(offset))
        pos, heading = _center_pose_at_point(lane, coord)
        out.append(PlannedMission((float(pos[0]), float(pos[1])), float(heading), (0.0, 0.0, 0.0), ()))
    return out


def load_missions(source: Union[str, dict]) -> Dict[str, Mission]:
    """Missions of a scenario as JSON (the ``missions.pkl`` of the reference's ``scenario build`` holds pickled
    sstudio objects): ``{agent id: {"begin": [road, lane index, offset], "end": [...], "via": [road, ...]}}``."""
    if isinstance(source, str):
        with open(source) as f:
            source = json.load(f)
    out = {}
    for agent_id, spec in source.items():
        out[agent_id] = Mission(Route(begin=tuple(spec["begin"]), end=tuple(spec["end"]), via=tuple(spec.get("via", ()))))
    return out
