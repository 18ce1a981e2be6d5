"""Oracle: SUMO road map, lanepoints and waypoint paths (test infrastructure).

Restates reference ``smarts/core/lanepoints.py``, the query half of
``smarts/core/sumo_road_network.py`` and the ``RoadMap.Lane`` reference methods of
``smarts/core/road_map.py:346-396`` on top of the parsed network
(:mod:`smarts_amd.sumo_map`, which stands in for the absent ``sumolib``).
KD-trees are ``scipy.spatial.KDTree`` exactly as in the reference
(``lanepoints.py:369-374``).
"""
import math
import queue
from collections import defaultdict

import numpy as np
from scipy.spatial import KDTree

from . import ref_math as rm


class OLane:
    """``SumoRoadNetwork.Lane`` (sumo_road_network.py:261-532), the queried subset."""

    def __init__(self, sumo_lane, omap):
        self._sl = sumo_lane
        self._map = omap
        self.lane_id = sumo_lane.getID()
        self.index = sumo_lane.getIndex()
        self._width = sumo_lane.getWidth()
        self.speed_limit = sumo_lane.getSpeed()
        self.length = sumo_lane.getLength()
        self.shape = sumo_lane.getShape(False)

    @property
    def road(self):
        return self._map.road_by_id(self._sl.getEdge().getID())

    @property
    def in_junction(self):
        return self.road.is_junction

    @property
    def outgoing_lanes(self):
        """sumo_road_network.py:350-358: via lane if present else the to-lane."""
        return [
            self._map.lane_by_id(c.getViaLaneID() or c.getToLane().getID())
            for c in self._sl.getOutgoing()
        ]

    @property
    def incoming_lanes(self):
        return [self._map.lane_by_id(l.getID()) for l in self._sl.getIncoming()]

    def width_at_offset(self, offset):
        return self._width

    def offset_along_lane(self, world_point):
        """sumo_road_network.py:476-491."""
        shape = self.shape
        point = tuple(world_point[:2])
        if point not in shape:
            return rm.polygon_offset_with_minimum_distance_to_point(point, shape)
        offset = 0
        for i in range(len(shape) - 1):
            if shape[i] == point:
                break
            offset += rm.euclidean_distance(shape[i], shape[i + 1])
        return offset

    def from_lane_coord(self, s):
        """sumo_road_network.py:502-506 (only the ``s`` coordinate is used)."""
        x, y = rm.position_at_shape_offset(self.shape, s)
        return (x, y, 0)

    def vector_at_offset(self, start_offset):
        """road_map.py:377-388."""
        if start_offset >= self.length:
            s_offset = self.length - 1
            end_offset = self.length
        else:
            s_offset = start_offset
            end_offset = start_offset + 1
        s_offset = max(s_offset, 0)
        p1 = self.from_lane_coord(s_offset)
        p2 = self.from_lane_coord(end_offset)
        return np.array(p2) - np.array(p1)

    def to_lane_coord(self, world_point):
        """road_map.py:346-355 -> (s, t)."""
        s = self.offset_along_lane(world_point)
        vector = self.vector_at_offset(s)
        normal = np.array([-vector[1], vector[0], 0])
        center_at_s = self.from_lane_coord(s)
        offcenter_vector = np.array(world_point) - center_at_s
        t_sign = np.sign(np.dot(offcenter_vector, normal))
        t = np.linalg.norm(offcenter_vector) * t_sign
        return s, t

    def oncoming_lanes_at_offset(self, offset):
        """sumo_road_network.py:371-395."""
        result = []
        radius = 1.1 * self.width_at_offset(offset)
        pt = self.from_lane_coord(offset)
        nearby_lanes = self._map.nearest_lanes(pt, radius=radius)
        if not nearby_lanes:
            return result
        my_vect = self.vector_at_offset(offset)
        my_norm = np.linalg.norm(my_vect)
        if my_norm == 0:
            return result
        threshold = -0.995562  # cos(175*pi/180)
        for lane, _ in nearby_lanes:
            if lane is self:
                continue
            s, _t = lane.to_lane_coord(pt)
            lv = lane.vector_at_offset(s)
            lv_norm = np.linalg.norm(lv)
            if lv_norm == 0:
                continue
            lane_angle = np.dot(my_vect, lv) / (my_norm * lv_norm)
            if lane_angle < threshold:
                result.append(lane)
        return result

    def center_pose_heading_at_point(self, point):
        """Heading of ``center_pose_at_point`` (road_map.py:390-396)."""
        offset = self.offset_along_lane(point)
        desired_vector = self.vector_at_offset(offset)
        return rm.pose_heading_from_angle(rm.vec_to_radians(desired_vector[:2]))


class ORoad:
    """``SumoRoadNetwork.Road`` (sumo_road_network.py:550-657), the queried subset."""

    def __init__(self, sumo_edge, omap):
        self._se = sumo_edge
        self._map = omap
        self.road_id = sumo_edge.getID()
        self.is_junction = sumo_edge.isSpecial()
        self.length = sumo_edge.getLength()

    @property
    def lanes(self):
        return [self._map.lane_by_id(l.getID()) for l in self._se.getLanes()]

    @property
    def outgoing_roads(self):
        return [self._map.road_by_id(e.getID()) for e in self._se.getOutgoing().keys()]

    @property
    def parallel_roads(self):
        """sumo_road_network.py:607-618 (a junction-internal edge has no from / to node in sumolib: the
        reference cannot be asked there; [] here)."""
        from_node, to_node = self._se.getFromNode(), self._se.getToNode()
        if from_node is None or to_node is None:
            return []
        return [
            self._map.road_by_id(edge.getID())
            for edge in from_node.getOutgoing()
            if self.road_id != edge.getID() and edge.getToNode().getID() == to_node.getID()
        ]

    def oncoming_roads_at_point(self, point):
        """sumo_road_network.py:596-605."""
        result = []
        for lane in self.lanes:
            offset = lane.to_lane_coord(point)[0]
            result += [ol.road for ol in lane.oncoming_lanes_at_offset(offset) if ol.road is not self]
        return result


class LP:
    """LinkedLanePoint + LanePoint flattened (lanepoints.py:46-70)."""

    __slots__ = ("lane", "pos", "orientation", "_heading", "is_inferred", "nexts", "idx")

    def __init__(self, lane, pos, orientation, is_inferred):
        self.lane = lane
        self.pos = np.asarray(pos, dtype=np.float64)
        self.orientation = orientation
        self._heading = None
        self.is_inferred = is_inferred
        self.nexts = []
        self.idx = -1

    @property
    def heading(self):
        # Pose.heading (coordinates.py:394-403)
        if self._heading is None:
            self._heading = rm.wrap_heading(rm.yaw_from_quat(self.orientation))
        return self._heading

    def key(self):
        # LanePoint equality = (lane, pose position+orientation, width) (lanepoints.py:46-55,
        # coordinates.py:272-280)
        return (self.lane.lane_id, self.pos[0], self.pos[1], self.orientation[2], self.orientation[3])


def shape_lanepoints(omap):
    """``LanePoints.from_sumo`` (lanepoints.py:104-227): shape points of every lane,
    linked across lanes through ``getOutgoing()`` / via lanes, breadth first."""
    net = omap.net
    memo = {}
    shape_lps = []

    def along_lane(lane):
        q = queue.Queue()
        q.put((lane, None))
        out = []
        while not q.empty():
            lane, previous_lp = q.get()
            first = memo.get(lane.getID())
            if first:
                if previous_lp:
                    previous_lp.nexts.append(first)
                continue
            lane_shape = [np.array(p) for p in lane.getShape(False)]
            assert len(lane_shape) >= 2
            olane = omap.lane_by_id(lane.getID())
            heading = rm.wrap_heading(rm.vec_to_radians(lane_shape[1] - lane_shape[0]))
            first = LP(olane, lane_shape[0], rm.quat_from_angle(heading), False)
            if previous_lp is not None:
                previous_lp.nexts.append(first)
            memo[lane.getID()] = first
            out.append(first)
            curr = first
            for p1, p2 in zip(lane_shape[1:], lane_shape[2:]):
                heading_ = rm.wrap_heading(rm.vec_to_radians(p2 - p1))
                llp = LP(olane, p1, rm.quat_from_angle(heading_), False)
                out.append(llp)
                curr.nexts.append(llp)
                curr = llp
            last = LP(olane, lane_shape[-1], curr.orientation, False)
            out.append(last)
            curr.nexts.append(last)
            curr = last
            for conn in lane.getOutgoing():
                out_lane = conn.getToLane()
                via = conn.getViaLaneID()
                if via:
                    out_lane = net.getLane(via)
                q.put((out_lane, curr))
        return out

    for edge in net.getEdges(False):
        for lane in edge.getLanes():
            shape_lps += along_lane(lane)
    return shape_lps


def interpolate_shape_lanepoints(shape_lps, spacing):
    """lanepoints.py:376-515."""
    interp_memo = {}
    linked = []

    def process_interp(shape_lp, first_linked, next_shape_lp, new_lps):
        # lanepoints.py:443-515
        rmlane = shape_lp.lane
        curr = first_linked
        lane_seg_vec = next_shape_lp.pos[:2] - shape_lp.pos[:2]
        lane_seg_len = np.linalg.norm(lane_seg_vec)
        dist_into_lane_seg = spacing
        while dist_into_lane_seg < lane_seg_len:
            p = dist_into_lane_seg / lane_seg_len
            pos = shape_lp.pos[:2] + lane_seg_vec * p
            last_spacing_threshold_dist = 0.8 * spacing
            minimum_dist_next_shape_lp = 1.4
            half_dist = np.linalg.norm(0.5 * (curr.pos[:2] - next_shape_lp.pos[:2]))
            mid_point = 0.5 * (next_shape_lp.pos[:2] + curr.pos[:2])
            if half_dist < minimum_dist_next_shape_lp:
                pos = mid_point
            dist_pos_next = np.linalg.norm(next_shape_lp.pos[:2] - pos)
            if dist_pos_next < last_spacing_threshold_dist:
                break
            heading = rm.vec_to_radians(lane_seg_vec)
            llp = LP(rmlane, pos, rm.quat_from_angle(heading), True)
            curr.nexts.append(llp)
            curr = llp
            new_lps.append(llp)
            dist_into_lane_seg += spacing
        return curr

    def from_shape_lp(shape_lp):
        # lanepoints.py:393-441
        q = queue.Queue()
        q.put((shape_lp, None))
        new_lps = []
        while not q.empty():
            shape_lp, previous_lp = q.get()
            first_linked = interp_memo.get(shape_lp.key())
            if first_linked:
                if previous_lp:
                    previous_lp.nexts.append(first_linked)
                continue
            first_linked = LP(shape_lp.lane, shape_lp.pos, shape_lp.orientation, False)
            if previous_lp is not None:
                previous_lp.nexts.append(first_linked)
            interp_memo[shape_lp.key()] = first_linked
            new_lps.append(first_linked)
            for current_shape_lp in shape_lp.nexts:
                if (
                    current_shape_lp.lane.lane_id == shape_lp.lane.lane_id
                    or current_shape_lp.lane in shape_lp.lane.outgoing_lanes
                ):
                    last_new = process_interp(shape_lp, first_linked, current_shape_lp, new_lps)
                    q.put((current_shape_lp, last_new))
                else:
                    q.put((current_shape_lp, first_linked))
        return new_lps

    for shape_lp in shape_lps:
        linked += from_shape_lp(shape_lp)
    for i, llp in enumerate(linked):
        llp.idx = i
    return linked


def _kd(lps):
    return KDTree(np.array([l.pos[:2] for l in lps]), leafsize=50)


class OLanePoints:
    """``LanePoints`` (lanepoints.py:73-102, 517-692).

    ``tie_rule``: the reference hands ties between *exactly equidistant* lanepoints
    (coincident lane-end / lane-start points on collinear lanes) to scipy's KD-tree,
    whose order among equal distances is an artefact of its heap and tree layout.
    ``"kdtree"`` keeps that behaviour (used to pin this restatement against the
    reference's own outputs); ``"index"`` orders equal distances by lanepoint index
    in the reference's global lanepoint order, which is the rule the device kernels
    implement (DESIGN.md "Deviations").  The two rules differ only on exact ties.
    """

    tie_rule = "index"

    def __init__(self, omap, spacing):
        self.linked = interpolate_shape_lanepoints(shape_lanepoints(omap), spacing)
        self.tree = _kd(self.linked)
        self.by_lane = defaultdict(list)
        self.by_road = defaultdict(list)
        for llp in self.linked:
            self.by_lane[llp.lane.lane_id].append(llp)
            self.by_road[llp.lane.road.road_id].append(llp)
        self.tree_by_lane = {k: _kd(v) for k, v in self.by_lane.items()}
        self.tree_by_road = {k: _kd(v) for k, v in self.by_road.items()}

    @classmethod
    def _closest_batched(cls, points, lps, tree, k=1):
        # lanepoints.py:517-524
        p2ds = np.array([np.array(p[:2]) for p in points])
        kk = min(k, len(lps))
        if cls.tie_rule == "kdtree":
            _, closest_indices = tree.query(p2ds, k=kk)
            closest_indices = np.atleast_2d(closest_indices)
            return [[lps[idx] for idx in idxs] for idxs in closest_indices]
        # same neighbours, equal distances ordered by global lanepoint index
        extra = min(len(lps), kk + 8)
        _, cand = tree.query(p2ds, k=extra)
        cand = np.asarray(cand).reshape(len(p2ds), -1)
        out = []
        for p, idxs in zip(p2ds, cand):
            keyed = []
            for i in idxs:
                q = lps[i].pos
                dx, dy = q[0] - p[0], q[1] - p[1]
                keyed.append((dx * dx + dy * dy, lps[i].idx, i))
            keyed.sort()
            out.append([lps[i] for _, _, i in keyed[:kk]])
        return out

    @staticmethod
    def _closest_with_pose(pos2d, heading, lps, tree, within_radius, k=10):
        # lanepoints.py:526-590 for one pose
        cands = OLanePoints._closest_batched([pos2d], lps, tree, k=k)[0]

        def sq(l):
            d = l.pos[:2] - pos2d
            return np.dot(d, d)

        cands = sorted(cands, key=sq)  # stable: equal distances keep the order above
        if within_radius is not None:
            radius_sq = within_radius * within_radius
            cands = [l for i, l in enumerate(cands) if sq(l) <= radius_sq or i == 0]
        return sorted(cands, key=lambda l: sq(l) + abs(rm.heading_relative_to(heading, l.heading)))

    def closest_lanepoint(self, pos2d, heading, within_radius=10, maximum_count=10):
        """``closest_lanepoints([pose])[0]`` (lanepoints.py:592-623)."""
        pos2d = np.asarray(pos2d[:2], dtype=np.float64)
        return self._closest_with_pose(pos2d, heading, self.linked, self.tree, within_radius, maximum_count)[0]

    def closest_linked_lanepoint_on_lane_to_point(self, point, lane_id):
        return self._closest_batched([point], self.by_lane[lane_id], self.tree_by_lane[lane_id], k=1)[0][0]

    def closest_linked_lanepoint_on_road(self, point, road_id):
        return self._closest_batched([point], self.by_road[road_id], self.tree_by_road[road_id])[0][0]

    def paths_starting_at_lanepoint(self, lanepoint, lookahead, filter_edge_ids):
        """lanepoints.py:646-692."""
        lanepoint_paths = [[lanepoint]]
        for _ in range(lookahead):
            next_paths = []
            for path in lanepoint_paths:
                branching = []
                for next_lp in path[-1].nexts:
                    next_lane = next_lp.lane
                    edge_id = next_lane.road.road_id
                    if filter_edge_ids and edge_id not in filter_edge_ids:
                        continue
                    if (
                        filter_edge_ids
                        and edge_id != filter_edge_ids[-1]
                        and all(ol.road.road_id not in filter_edge_ids for ol in next_lane.outgoing_lanes)
                    ):
                        continue
                    branching.append(path + [next_lp])
                if not branching:
                    branching = [path]
                next_paths += branching
            lanepoint_paths = next_paths
        return lanepoint_paths


class Waypoint:
    """road_map.py:556-618."""

    __slots__ = ("pos", "heading", "lane_id", "lane_width", "speed_limit", "lane_index")

    def __init__(self, pos, heading, lane_id, lane_width, speed_limit, lane_index):
        self.pos = pos
        self.heading = heading
        self.lane_id = lane_id
        self.lane_width = lane_width
        self.speed_limit = speed_limit
        self.lane_index = lane_index

    def relative_heading(self, h):
        return rm.heading_relative_to(self.heading, h)

    def signed_lateral_error(self, p):
        return rm.signed_dist_to_line(p, self.pos, rm.radians_to_vec(self.heading))

    def dist_to(self, p):
        return np.linalg.norm(self.pos - np.asarray(p)[: len(self.pos)])


def equally_spaced_path(path, point, lp_spacing):
    """``SumoRoadNetwork._equally_spaced_path`` (sumo_road_network.py:1312-1437)."""
    cont = ["positions_x", "positions_y", "headings", "lane_width", "speed_limit"]
    disc = ["lane_id", "lane_index"]
    ref = {k: [] for k in cont + disc}
    for idx, lanepoint in enumerate(path):
        if lanepoint.is_inferred and 0 < idx < len(path) - 1:
            continue
        ref["positions_x"].append(lanepoint.pos[0])
        ref["positions_y"].append(lanepoint.pos[1])
        ref["headings"].append(lanepoint.heading)
        ref["lane_id"].append(lanepoint.lane.lane_id)
        ref["lane_index"].append(lanepoint.lane.index)
        ref["lane_width"].append(lanepoint.lane._width)
        ref["speed_limit"].append(lanepoint.lane.speed_limit)

    ref["headings"] = rm.inplace_unwrap(ref["headings"])
    first_lp_heading = ref["headings"][0]
    lp_position = path[0].pos[:2]
    vehicle_pos = np.array(point[:2])
    heading_vec = np.array(rm.radians_to_vec(first_lp_heading))
    projected = np.inner((vehicle_pos - lp_position), heading_vec)
    ref["positions_x"][0] = lp_position[0] + projected * heading_vec[0]
    ref["positions_y"][0] = lp_position[1] + projected * heading_vec[1]

    cumulative = np.cumsum(
        np.sqrt(
            np.ediff1d(ref["positions_x"], to_begin=0) ** 2
            + np.ediff1d(ref["positions_y"], to_begin=0) ** 2
        )
    )
    if len(cumulative) <= lp_spacing:
        lp = path[0]
        return [Waypoint(lp.pos[:2], lp.heading, lp.lane.lane_id, lp.lane._width, lp.lane.speed_limit, lp.lane.index)]

    even = np.linspace(0, cumulative[-1], len(path))
    evenly = {}
    for variable in cont:
        evenly[variable] = np.interp(even, cumulative, ref[variable])
    for variable in disc:
        ref_coordinates = ref[variable]
        evenly[variable] = []
        jdx = 0
        for idx in range(len(path)):
            while jdx + 1 < len(cumulative) and even[idx] > cumulative[jdx + 1]:
                jdx += 1
            evenly[variable].append(ref_coordinates[jdx])
        evenly[variable].append(ref_coordinates[-1])

    out = []
    for idx in range(len(path)):
        out.append(
            Waypoint(
                pos=np.array([evenly["positions_x"][idx], evenly["positions_y"][idx]]),
                heading=rm.wrap_heading(evenly["headings"][idx]),
                lane_width=evenly["lane_width"][idx],
                speed_limit=evenly["speed_limit"][idx],
                lane_id=evenly["lane_id"][idx],
                lane_index=evenly["lane_index"][idx],
            )
        )
    return out


class ORoadNetwork:
    """``SumoRoadNetwork`` (sumo_road_network.py), query side."""

    def __init__(self, net, lanepoint_spacing=1.0, default_lane_width=3.2):
        self.net = net
        self._default_lane_width = default_lane_width
        self._spacing = lanepoint_spacing
        self._lanes = {}
        self._roads = {}
        self._all_lanes = net.all_lanes()
        self._lane_bbox = [l.getBoundingBox(False) for l in self._all_lanes]
        self.lanepoints = OLanePoints(self, lanepoint_spacing)

    def lane_by_id(self, lane_id):
        lane = self._lanes.get(lane_id)
        if lane is None:
            lane = OLane(self.net.getLane(lane_id), self)
            self._lanes[lane_id] = lane
        return lane

    def lane_bands(self):
        """(centre-line shape, width) of every lane — what the road mesh is built from
        (sumo_road_network.py:986-1019 buffers each lane shape by half its width)."""
        if getattr(self, "_bands", None) is None:
            self._bands = [([(float(p[0]), float(p[1])) for p in sl.getShape(False)], float(sl.getWidth()))
                           for sl in self._all_lanes]
        return self._bands

    def road_by_id(self, road_id):
        road = self._roads.get(road_id)
        if road is None:
            road = ORoad(self.net.getEdge(road_id), self)
            self._roads[road_id] = road
        return road

    # ---- nearest lanes ----
    def nearest_lanes(self, point, radius=None, include_junctions=True):
        """sumo_road_network.py:675-701.  ``getNeighboringLanes`` (sumolib, absent) keeps
        lanes whose bounding box meets the query square and whose centre polyline is
        closer than ``radius``; with ``include_junctions=True`` the plain lane shape is
        used (``includeJunctions=False``, the inversion noted at :682-688)."""
        if radius is None:
            radius = max(10, 2 * self._default_lane_width)
        x, y = point[0], point[1]
        with_junction_pos = not include_junctions
        cands = []
        for i, sl in enumerate(self._all_lanes):
            if with_junction_pos:
                bx0, by0, bx1, by1 = sl.getBoundingBox(True)
            else:
                bx0, by0, bx1, by1 = self._lane_bbox[i]
            if bx1 < x - radius or bx0 > x + radius or by1 < y - radius or by0 > y + radius:
                continue
            d = rm.distance_point_to_polygon((x, y), sl.getShape(with_junction_pos))
            if d < radius:
                cands.append((sl, d))
        if not include_junctions:
            cands = [c for c in cands if not c[0].getEdge().isSpecial()]
        cands.sort(key=lambda t: t[1])
        return [(self.lane_by_id(sl.getID()), d) for sl, d in cands]

    def nearest_lane(self, point, radius=None, include_junctions=True):
        """road_map.py:91-96."""
        nl = self.nearest_lanes(point, radius, include_junctions)
        return nl[0][0] if nl else None

    def road_with_point(self, point):
        """sumo_road_network.py:703-709."""
        radius = max(5, 2 * self._default_lane_width)
        for nl, dist in self.nearest_lanes(point, radius):
            if dist < 0.5 * nl._width + 1e-1:
                return nl.road
        return None

    # ---- routes ----
    def generate_routes(self, start_road, end_road, via=()):
        """sumo_road_network.py:711-765 -> the route's road ids ([] = none found).  The edge search is
        ``sumolib``'s ``getShortestPath`` (absent; restated in smarts_amd.sumo_map.SumoNet)."""
        roads = [start_road] + list(via)
        if end_road is not start_road:
            roads.append(end_road)
        edges = []
        for cur_road, next_road in zip(roads, roads[1:] + [None]):
            if not next_road:
                edges.append(cur_road._se)
                break
            sub_route = self.net.getShortestPath(cur_road._se, next_road._se)[0] or []
            if len(sub_route) < 2:
                return []
            edges.extend(sub_route[:-1])
        if len(edges) == 1:
            return [edges[0].getID()]
        used_edges = []
        edge_ids = []
        for cur_edge, next_edge in zip(edges, edges[1:]):
            for internal_route in self._internal_routes_between(cur_edge, next_edge):
                used_edges.extend(internal_route)
                edge_ids.extend([edge.getID() for edge in internal_route])
        _, indices = np.unique(edge_ids, return_index=True)
        return [used_edges[idx].getID() for idx in sorted(indices)]

    def _internal_routes_between(self, start_edge, end_edge):
        """sumo_road_network.py:767-800."""
        routes = []
        outgoing = start_edge.getOutgoing()
        assert end_edge in outgoing
        for connection in outgoing[end_edge]:
            conn_route = [start_edge]
            via_lane_id = connection.getViaLaneID()
            while via_lane_id:
                via_edge = self.lane_by_id(via_lane_id).road._se
                conn_route.append(via_edge)
                next_via_lane_ids = set(conn.getViaLaneID() for conn in via_edge.getOutgoing()[end_edge])
                assert len(next_via_lane_ids) == 1
                via_lane_id = list(next_via_lane_ids)[0]
            conn_route.append(end_edge)
            routes.append(conn_route)
        return routes

    def create_route(self, start_point, goal_point, route_vias=()):
        """``Plan.create_route`` for a fixed-route mission (plan.py:316-349) -> road ids."""
        start_lane = self.nearest_lane(start_point, include_junctions=False)
        assert start_lane, "route must start in a lane"
        end_lane = self.nearest_lane(goal_point, include_junctions=False)
        assert end_lane, "route must end in a lane"
        via_roads = [self.road_by_id(v) for v in route_vias]
        return self.generate_routes(start_lane.road, end_lane.road, via_roads)

    # ---- waypoint paths ----
    def _waypoints_starting_at_lanepoint(self, lanepoint, lookahead, filter_road_ids, point):
        """sumo_road_network.py:1278-1310 without the shared cache: the cache returns
        a pure function of its key (Appendix A #7 of SURVEY.md)."""
        paths = self.lanepoints.paths_starting_at_lanepoint(lanepoint, lookahead, filter_road_ids)
        return [equally_spaced_path(p, point, self._spacing) for p in paths]

    def lane_waypoint_paths_at(self, lane, point, lookahead, filter_road_ids=None):
        """``Lane._waypoint_paths_at`` (sumo_road_network.py:429-446)."""
        llp = self.lanepoints.closest_linked_lanepoint_on_lane_to_point(point, lane.lane_id)
        return self._waypoints_starting_at_lanepoint(
            llp, lookahead, tuple(filter_road_ids) if filter_road_ids else (), tuple(point)
        )

    def _resolve_in_junction(self, position, heading):
        """sumo_road_network.py:842-860."""
        lp = self.lanepoints.closest_lanepoint(position, heading, within_radius=None)
        lane = lp.lane
        if not lane.in_junction:
            return []
        road_ids = [lane.road.road_id]
        next_roads = lane.road.outgoing_roads
        assert len(next_roads) <= 1
        if next_roads:
            road_ids.append(next_roads[0].road_id)
        return road_ids

    def _waypoint_paths_along_route(self, point, lookahead, route):
        """sumo_road_network.py:862-882."""
        cands = [self.lanepoints.closest_linked_lanepoint_on_road(point, road) for road in route]
        closest = min(cands, key=lambda l: np.linalg.norm(l.pos[:2] - np.array(point[:2])))
        paths = []
        for lane in closest.lane.road.lanes:
            paths += self.lane_waypoint_paths_at(lane, point, lookahead, route)
        return sorted(paths, key=lambda p: p[0].lane_index)

    def waypoint_paths(self, position, heading, lookahead, within_radius=5, route=None):
        """sumo_road_network.py:815-840.  ``route`` is ``None`` (no route object), or a
        list of road ids (possibly empty = the endless-mission ``empty_route()``)."""
        if route is not None:
            road_ids = list(route) if route else self._resolve_in_junction(position, heading)
            if road_ids:
                return self._waypoint_paths_along_route(position, lookahead, road_ids)
        closest_lane = self.lanepoints.closest_lanepoint(position, heading, within_radius=within_radius).lane
        paths = []
        for lane in closest_lane.road.lanes:
            paths += self.lane_waypoint_paths_at(lane, position, lookahead)
        return sorted(paths, key=lambda p: p[0].lane_index)
