"""SUMO road-network loader (host side).

The reference reaches its map through ``sumolib`` (eclipse-sumo==1.10.0, reference
``setup.py:28``), which is *not* vendored under the reference tree.  This module
restates the small part of ``sumolib.net`` that the hot path consumes, from the
reference's own call sites:

* ``sumolib.net.readNet(net_file, withInternal=True)``
  (reference ``smarts/core/sumo_road_network.py:154``)
* ``lane.getShape(False)``, ``lane.getOutgoing()``, ``conn.getViaLaneID()``,
  ``conn.getToLane()`` (reference ``smarts/core/lanepoints.py:136,202-210``)
* ``lane.getSpeed/getLength/getWidth/getIndex`` (``sumo_road_network.py:279-299``)
* ``edge.isSpecial/getLanes/getOutgoing`` (``sumo_road_network.py:559-625``)
* ``net.getBoundary()`` and the origin shift that ``scl scenario build`` applies
  through ``netconvert --offset.disable-normalization=FALSE``
  (``sumo_road_network.py:104-176``)
* the polyline helpers of ``sumolib.geomhelper`` (readable in-tree twins: reference
  ``smarts/core/utils/math.py:293-433``), here in vectorised form.

Nothing here touches the GPU; the output feeds :mod:`smarts_amd.map_compiler`.
"""
from __future__ import annotations

import gzip
import json
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

DEFAULT_LANE_WIDTH = 3.2  # reference sumo_road_network.py:80 / sumolib default

Point2 = Tuple[float, float]


# --------------------------------------------------------------------------------------
# polyline geometry (vectorised; semantics of sumolib.geomhelper.distancePointToPolygon /
# positionAtShapeOffset as used by reference sumo_road_network.py:481-506,689-695)
# --------------------------------------------------------------------------------------
def polyline_distance(point: Sequence[float], shape: np.ndarray) -> float:
    """Minimum distance from ``point`` to the polyline ``shape`` ((n, 2) array).

    Per segment this is the clamped-projection distance: the foot of the
    perpendicular when it falls inside the segment, else the nearer end point.
    """
    a = shape[:-1]
    b = shape[1:]
    ab = b - a
    seg_len = np.sqrt(ab[:, 0] * ab[:, 0] + ab[:, 1] * ab[:, 1])
    u = (point[0] - a[:, 0]) * ab[:, 0] + (point[1] - a[:, 1]) * ab[:, 1]
    ll = seg_len * seg_len
    before = (seg_len == 0.0) | (u < 0.0)
    after = ~before & (u > ll)
    with np.errstate(divide="ignore", invalid="ignore"):
        off = np.where(before, 0.0, np.where(after, seg_len, u / seg_len))
        t = np.where(off == 0.0, 0.0, off / seg_len)
    fx = a[:, 0] + t * ab[:, 0]
    fy = a[:, 1] + t * ab[:, 1]
    d = np.sqrt((point[0] - fx) ** 2 + (point[1] - fy) ** 2)
    return float(d.min())


def polyline_point_at(shape: np.ndarray, offset: float) -> Tuple[float, float]:
    """Point ``offset`` metres along the polyline (clamped to its last vertex)."""
    seg = np.sqrt(((shape[1:] - shape[:-1]) ** 2).sum(axis=1))
    cum = np.concatenate(([0.0], np.cumsum(seg)))
    if offset >= cum[-1]:
        return (float(shape[-1, 0]), float(shape[-1, 1]))
    i = int(np.searchsorted(cum, offset, side="right") - 1)
    i = min(max(i, 0), len(seg) - 1)
    rem = offset - cum[i]
    if seg[i] == 0.0 or rem <= 0.0:
        return (float(shape[i, 0]), float(shape[i, 1]))
    f = rem / seg[i]
    return (
        float(shape[i, 0] + (shape[i + 1, 0] - shape[i, 0]) * f),
        float(shape[i, 1] + (shape[i + 1, 1] - shape[i, 1]) * f),
    )


# --------------------------------------------------------------------------------------
# network objects
# --------------------------------------------------------------------------------------
@dataclass
class Connection:
    """One ``<connection>`` element (sumolib.net.connection.Connection)."""

    from_edge: "Edge"
    to_edge: "Edge"
    from_lane: "Lane"
    to_lane: "Lane"
    via_lane_id: str
    direction: str = ""
    state: str = ""

    def getViaLaneID(self) -> str:
        return self.via_lane_id

    def getToLane(self) -> "Lane":
        return self.to_lane

    def getFromLane(self) -> "Lane":
        return self.from_lane


@dataclass(eq=False)
class Lane:
    """A SUMO lane with the accessors the reference uses."""

    lane_id: str
    edge: "Edge"
    index: int
    speed: float
    length: float
    width: float
    shape: List[Point2]
    outgoing: List[Connection] = field(default_factory=list)

    def getID(self) -> str:
        return self.lane_id

    def getEdge(self) -> "Edge":
        return self.edge

    def getIndex(self) -> int:
        return self.index

    def getSpeed(self) -> float:
        return self.speed

    def getLength(self) -> float:
        return self.length

    def getWidth(self) -> float:
        return self.width

    def getShape(self, includeJunctions: bool = False) -> List[Point2]:
        if includeJunctions and not self.edge.isSpecial():
            return self.edge.net._shape_with_junctions(self)
        return self.shape

    def getOutgoing(self) -> List[Connection]:
        return self.outgoing

    def getIncoming(self) -> List["Lane"]:
        # sumolib: lanes with a connection into this lane (normal and internal edges).
        lanes = []
        for conns in self.edge.incoming.values():
            for c in conns:
                if c.to_lane is self:
                    lanes.append(c.from_lane)
        return lanes

    def getBoundingBox(self, includeJunctions: bool = False):
        s = self.getShape(includeJunctions)
        xs = [p[0] for p in s]
        ys = [p[1] for p in s]
        return (min(xs), min(ys), max(xs), max(ys))


@dataclass(eq=False)
class Edge:
    edge_id: str
    function: str
    from_node: Optional[str]
    to_node: Optional[str]
    net: "SumoNet"
    lanes: List[Lane] = field(default_factory=list)
    outgoing: Dict["Edge", List[Connection]] = field(default_factory=dict)
    incoming: Dict["Edge", List[Connection]] = field(default_factory=dict)

    def getID(self) -> str:
        return self.edge_id

    def isSpecial(self) -> bool:
        return self.function != ""

    def getFunction(self) -> str:
        return self.function

    def getLanes(self) -> List[Lane]:
        return self.lanes

    def getLane(self, idx: int) -> Lane:
        return self.lanes[idx]

    def getLength(self) -> float:
        return self.lanes[0].length

    def getOutgoing(self) -> Dict["Edge", List[Connection]]:
        return self.outgoing

    def getIncoming(self) -> Dict["Edge", List[Connection]]:
        return self.incoming

    def getFromNode(self) -> Optional["Node"]:
        """sumolib: the junction the edge leaves (None for junction-internal edges, which carry no from / to)."""
        return self.net.node_objects.get(self.from_node) if self.from_node else None

    def getToNode(self) -> Optional["Node"]:
        return self.net.node_objects.get(self.to_node) if self.to_node else None

    def __hash__(self):
        return id(self)


@dataclass(eq=False)
class Node:
    """``sumolib.net.node.Node``: id + the edges that leave / enter it, in the net file's edge order."""

    node_id: str
    outgoing: List[Edge] = field(default_factory=list)
    incoming: List[Edge] = field(default_factory=list)

    def getID(self) -> str:
        return self.node_id

    def getOutgoing(self) -> List[Edge]:
        return self.outgoing

    def getIncoming(self) -> List[Edge]:
        return self.incoming


class SumoNet:
    """The subset of ``sumolib.net.Net`` consumed by the hot path."""

    def __init__(self):
        self.edges: List[Edge] = []
        self.id2edge: Dict[str, Edge] = {}
        self.id2lane: Dict[str, Lane] = {}
        self.nodes: Dict[str, Point2] = {}
        self.node_objects: Dict[str, Node] = {}
        self.conv_boundary: Tuple[float, float, float, float] = (0.0, 0.0, 0.0, 0.0)
        self.shifted_by: Tuple[float, float] = (0.0, 0.0)
        self.source: str = ""

    # ---- sumolib-like accessors ----
    def getBoundary(self):
        return list(self.conv_boundary)

    def getEdges(self, withInternal: bool = True) -> List[Edge]:
        if withInternal:
            return self.edges
        return [e for e in self.edges if e.function == ""]

    def getEdge(self, edge_id: str) -> Optional[Edge]:
        return self.id2edge.get(edge_id)

    def getLane(self, lane_id: str) -> Optional[Lane]:
        return self.id2lane.get(lane_id)

    def getShortestPath(self, fromEdge: Edge, toEdge: Edge, maxCost: float = float("inf")):
        """``sumolib.net.Net.getShortestPath`` (eclipse-sumo 1.10.0, absent here) for the arguments the
        reference passes (``sumo_road_network.py:733-737``: two edges, defaults otherwise): Dijkstra over
        the normal edges, an edge's cost = its length, the start edge's own length included; the heap
        holds (cost, edge id, ...) so equal costs leave in edge-id order.  Returns (edges, cost) or
        (None, inf).  Pinned by the reference's ``test_map.py:123-125`` (tests/test_missions.py)."""
        import heapq

        heap = [(fromEdge.getLength(), fromEdge.getID(), fromEdge, ())]
        seen = set()
        dist = {fromEdge: fromEdge.getLength()}
        while heap:
            cost, _, e1, path = heapq.heappop(heap)
            if e1 in seen:
                continue
            seen.add(e1)
            path = (e1, path)
            if e1 is toEdge:
                out = []
                while path:
                    out.append(path[0])
                    path = path[1]
                out.reverse()
                return out, cost
            if cost > maxCost:
                return None, cost
            for e2 in e1.getOutgoing():
                if e2.isSpecial() or e2 in seen:
                    continue
                new_cost = cost + e2.getLength()
                if e2 not in dist or new_cost < dist[e2]:
                    dist[e2] = new_cost
                    heapq.heappush(heap, (new_cost, e2.getID(), e2, path))
        return None, float("inf")

    def all_lanes(self) -> List[Lane]:
        """Lanes in sumolib ``_allLanes`` order (edge file order, then lane index)."""
        out = []
        for e in self.edges:
            out += e.lanes
        return out

    def _shape_with_junctions(self, lane: Lane) -> List[Point2]:
        # sumolib.net.lane.addJunctionPos: prepend/append the from/to node position
        # unless it coincides with the shape end.
        shape = list(lane.shape)
        fr = self.nodes.get(lane.edge.from_node)
        to = self.nodes.get(lane.edge.to_node)
        if fr is not None and fr != shape[0]:
            shape = [fr] + shape
        if to is not None and to != shape[-1]:
            shape = shape + [to]
        return shape

    # ---- queries restated from the reference's call sites ----
    def neighboring_lanes(self, x: float, y: float, r: float, include_junctions_shape: bool = False):
        """``Net.getNeighboringLanes(x, y, r, includeJunctions, allowFallback=False)``.

        sumolib intersects an R-tree of lane bounding boxes with the query square
        and keeps lanes whose polyline distance is ``< r``.  The R-tree only
        prunes, so the result set is ``{lane : dist(lane) < r}``; order here is
        ``_allLanes`` order (the caller sorts by distance, stably).
        """
        out = []
        for lane in self.all_lanes():
            bx0, by0, bx1, by1 = lane.getBoundingBox(include_junctions_shape)
            if bx1 < x - r or bx0 > x + r or by1 < y - r or by0 > y + r:
                continue
            d = polyline_distance((x, y), np.asarray(lane.getShape(include_junctions_shape), dtype=np.float64))
            if d < r:
                out.append((lane, d))
        return out

    # ---- (de)serialisation of the compact network description ----
    def to_dict(self) -> dict:
        return {
            "format": "smx-net-1",
            "conv_boundary": list(self.conv_boundary),
            "shifted_by": list(self.shifted_by),
            "nodes": {k: list(v) for k, v in self.nodes.items()},
            "edges": [
                {
                    "id": e.edge_id,
                    "function": e.function,
                    "from": e.from_node,
                    "to": e.to_node,
                    "lanes": [
                        {
                            "id": l.lane_id,
                            "index": l.index,
                            "speed": l.speed,
                            "length": l.length,
                            "width": l.width,
                            "shape": [list(p) for p in l.shape],
                        }
                        for l in e.lanes
                    ],
                }
                for e in self.edges
            ],
            "connections": [
                {
                    "from": c.from_edge.edge_id,
                    "to": c.to_edge.edge_id,
                    "fromLane": c.from_lane.index,
                    "toLane": c.to_lane.index,
                    "via": c.via_lane_id,
                    "dir": c.direction,
                    "state": c.state,
                }
                for c in self._connections
            ],
        }

    def save(self, path: str):
        data = json.dumps(self.to_dict(), separators=(",", ":")).encode()
        with gzip.GzipFile(path, "wb", mtime=0) as f:
            f.write(data)

    _connections: List[Connection]


def _add_connection(net: SumoNet, from_edge, to_edge, from_lane, to_lane, via, direction, state):
    conn = Connection(from_edge, to_edge, from_lane, to_lane, via, direction, state)
    from_edge.outgoing.setdefault(to_edge, []).append(conn)
    from_lane.outgoing.append(conn)
    to_edge.incoming.setdefault(from_edge, []).append(conn)
    net._connections.append(conn)
    if via:
        # sumolib also registers the implicit (from -> via) connection as incoming
        # of the internal edge so that internal lanes know their predecessors.
        via_lane = net.id2lane.get(via)
        if via_lane is not None:
            via_edge = via_lane.edge
            via_edge.incoming.setdefault(from_edge, []).append(
                Connection(from_edge, via_edge, from_lane, via_lane, "", direction, state)
            )


def _parse_shape(text: str) -> List[Point2]:
    pts = []
    for tok in text.split():
        c = tok.split(",")
        pts.append((float(c[0]), float(c[1])))
    return pts


def _shift_value(v: float, d: float) -> float:
    # netconvert writes 2-decimal coordinates; keep the shifted value on that grid so
    # that e.g. 145.20 - 20.00 parses exactly like the text "125.20".
    return float(f"{v + d:.2f}")


def _build(desc: dict, shift_to_origin: bool) -> SumoNet:
    net = SumoNet()
    net._connections = []
    bb = tuple(desc["conv_boundary"])
    dx = dy = 0.0
    already = tuple(desc.get("shifted_by", (0.0, 0.0)))
    origin_ok = bb[0] <= 0.0 and bb[1] <= 0.0 and bb[2] >= 0.0 and bb[3] >= 0.0
    if shift_to_origin and not origin_ok:
        # reference sumo_road_network.py:157-176: netconvert offset normalisation moves
        # the lower-left corner of the boundary to (0, 0).
        dx, dy = -bb[0], -bb[1]
        bb = (0.0, 0.0, _shift_value(bb[2], dx), _shift_value(bb[3], dy))
    net.conv_boundary = bb
    net.shifted_by = (already[0] + dx, already[1] + dy)

    def sh(p):
        if dx == 0.0 and dy == 0.0:
            return (float(p[0]), float(p[1]))
        return (_shift_value(p[0], dx), _shift_value(p[1], dy))

    for nid, xy in desc.get("nodes", {}).items():
        net.nodes[nid] = sh(xy)
    for ed in desc["edges"]:
        e = Edge(ed["id"], ed.get("function", "") or "", ed.get("from"), ed.get("to"), net)
        for ld in ed["lanes"]:
            lane = Lane(
                ld["id"],
                e,
                int(ld["index"]),
                float(ld["speed"]),
                float(ld["length"]),
                float(ld.get("width", DEFAULT_LANE_WIDTH)),
                [sh(p) for p in ld["shape"]],
            )
            e.lanes.append(lane)
            net.id2lane[lane.lane_id] = lane
        net.edges.append(e)
        net.id2edge[e.edge_id] = e
        if e.from_node:
            net.node_objects.setdefault(e.from_node, Node(e.from_node)).outgoing.append(e)
        if e.to_node:
            net.node_objects.setdefault(e.to_node, Node(e.to_node)).incoming.append(e)
    for cd in desc["connections"]:
        fe = net.id2edge.get(cd["from"])
        te = net.id2edge.get(cd["to"])
        if fe is None or te is None:
            continue
        _add_connection(
            net,
            fe,
            te,
            fe.lanes[int(cd["fromLane"])],
            te.lanes[int(cd["toLane"])],
            cd.get("via", "") or "",
            cd.get("dir", ""),
            cd.get("state", ""),
        )
    return net


def _desc_from_net_xml(path: str) -> dict:
    root = ET.parse(path).getroot()
    loc = root.find("location")
    desc = {
        "format": "smx-net-1",
        "conv_boundary": [float(v) for v in loc.get("convBoundary").split(",")],
        "shifted_by": [0.0, 0.0],
        "nodes": {},
        "edges": [],
        "connections": [],
    }
    for j in root.findall("junction"):
        if j.get("type") == "internal":
            continue
        desc["nodes"][j.get("id")] = [float(j.get("x")), float(j.get("y"))]
    for e in root.findall("edge"):
        function = e.get("function", "") or ""
        if function not in ("", "internal"):
            # crossings / walking areas carry no vehicle lanes on this path
            continue
        lanes = []
        for l in e.findall("lane"):
            lanes.append(
                {
                    "id": l.get("id"),
                    "index": int(l.get("index")),
                    "speed": float(l.get("speed")),
                    "length": float(l.get("length")),
                    "width": float(l.get("width", DEFAULT_LANE_WIDTH)),
                    "shape": [list(p) for p in _parse_shape(l.get("shape", ""))],
                }
            )
        desc["edges"].append(
            {"id": e.get("id"), "function": function, "from": e.get("from"), "to": e.get("to"), "lanes": lanes}
        )
    for c in root.findall("connection"):
        desc["connections"].append(
            {
                "from": c.get("from"),
                "to": c.get("to"),
                "fromLane": int(c.get("fromLane")),
                "toLane": int(c.get("toLane")),
                "via": c.get("via", "") or "",
                "dir": c.get("dir", ""),
                "state": c.get("state", ""),
            }
        )
    return desc


SMX_NET_NAME = "map.smxnet.json.gz"


def load_net(source: str, shift_to_origin: bool = True) -> SumoNet:
    """Load a SUMO network.

    ``source`` may be a scenario directory (``map.net.xml`` inside, as in reference
    ``sumo_road_network.py:192-196``, or the compact ``map.smxnet.json.gz`` written by
    ``tools/import_sumo_net.py``), a ``.net.xml`` file, or a compact file.

    ``shift_to_origin`` reproduces what ``scl scenario build`` does by default
    (reference ``cli/studio.py:74-76``): maps whose boundary does not contain the
    origin are translated so that it does.
    """
    path = source
    if os.path.isdir(source):
        xml_path = os.path.join(source, "map.net.xml")
        smx_path = os.path.join(source, SMX_NET_NAME)
        path = xml_path if os.path.isfile(xml_path) else smx_path
    if not os.path.isfile(path):
        raise FileNotFoundError(f"no SUMO network at {source!r}")
    if path.endswith(".xml"):
        desc = _desc_from_net_xml(path)
    else:
        with gzip.open(path, "rb") as f:
            desc = json.loads(f.read().decode())
        if desc.get("format") != "smx-net-1":
            raise ValueError(f"{path}: not an smx-net-1 file")
    net = _build(desc, shift_to_origin)
    net.source = path
    return net
