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
