"""Map compiler: SUMO network -> flat struct-of-arrays tables for the HIP kernels.

Load-time work on the host.  What the reference builds as
linked Python objects and KD-/R-trees, this builds as flat arrays + uniform grids:

* lane / road tables (``sumo_road_network.py:261-358,550-625``);
* the 1 m lanepoint graph of ``LanePoints.from_sumo`` +
  ``_interpolate_shape_lanepoints`` (``lanepoints.py:104-227,376-515``) in the
  reference's own global order, as CSR adjacency;
* a uniform grid over lanepoints (replaces the scipy KD-trees of
  ``lanepoints.py:369-374``) and one over lane centre-line segments (replaces the
  rtree behind ``getNeighboringLanes``, ``sumo_road_network.py:689-695``).

All trigonometry that feeds *tables* is done here with the same libm calls the
reference makes (``math.atan2/sin/cos``), so table values are bit-identical to the
reference's lanepoint poses; the kernels never re-derive them.
"""
from __future__ import annotations

import math
from collections import deque
from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np

from .sumo_map import SumoNet

TWO_PI = 2 * math.pi


def _vec_to_radians(x: float, y: float) -> float:
    # heading convention of the reference: 0 = +y, counter-clockwise, in [0, 2pi)
    # (smarts/core/utils/math.py:256-277)
    r = math.atan2(abs(y), abs(x))
    if x < 0:
        if y < 0:
            return (r + 0.5 * math.pi) % TWO_PI
        return (0.5 * math.pi - r) % TWO_PI
    elif y < 0:
        return (1.5 * math.pi - r) % TWO_PI
    return (r - 0.5 * math.pi) % TWO_PI


def _wrap(h: float) -> float:
    h = h % TWO_PI
    if h > math.pi:
        h -= TWO_PI
    return h


def _pose_heading(angle: float) -> float:
    """Heading as the reference stores it on a lanepoint: through the z-quaternion
    and back (coordinates.py:394-403, math.py:78-106)."""
    half = angle * 0.5
    z, w = math.sin(half), math.cos(half)
    # numpy arctan2 on float64 scalars == libm atan2
    return _wrap(float(np.arctan2(2 * (0 * 0 + w * z), w ** 2 + 0 ** 2 - 0 ** 2 - z ** 2)))


@dataclass
class CompiledMap:
    """Flat tables; every array is C-contiguous and ready for upload."""

    # lanes (index = position in sumolib ``_allLanes`` order)
    lane_ids: List[str]
    lane_road: np.ndarray
    lane_index: np.ndarray
    lane_width: np.ndarray
    lane_speed: np.ndarray
    lane_length: np.ndarray
    lane_in_junction: np.ndarray
    lane_shape_off: np.ndarray
    shape_x: np.ndarray
    shape_y: np.ndarray
    lane_out_off: np.ndarray
    lane_out_idx: np.ndarray
    lane_in_off: np.ndarray   # Lane.incoming_lanes (sumo_road_network.py:342-348), in sumolib's order
    lane_in_idx: np.ndarray
    # roads
    road_ids: List[str]
    road_lane_off: np.ndarray
    road_lanes: np.ndarray
    road_is_junction: np.ndarray
    road_out_road: np.ndarray
    road_par_off: np.ndarray  # Road.parallel_roads (sumo_road_network.py:607-618)
    road_par_idx: np.ndarray
    # lanepoints
    lp_x: np.ndarray
    lp_y: np.ndarray
    lp_heading: np.ndarray
    lp_dirx: np.ndarray
    lp_diry: np.ndarray
    lp_lane: np.ndarray
    lp_inferred: np.ndarray
    lp_next_off: np.ndarray
    lp_next_idx: np.ndarray
    # lanepoint grid
    lpg_origin: np.ndarray
    lpg_cell: float
    lpg_dims: np.ndarray
    lpg_off: np.ndarray
    lpg_idx: np.ndarray
    # segment grid
    seg_lane: np.ndarray
    seg_v0: np.ndarray
    sg_origin: np.ndarray
    sg_cell: float
    sg_dims: np.ndarray
    sg_off: np.ndarray
    sg_idx: np.ndarray
    # misc
    default_lane_width: float = 3.2
    lanepoint_spacing: float = 1.0
    max_fanout: int = 1
    shifted_by: tuple = (0.0, 0.0)
    extras: Dict[str, object] = field(default_factory=dict)

    @property
    def n_lanes(self) -> int:
        return len(self.lane_ids)

    @property
    def n_lanepoints(self) -> int:
        return len(self.lp_x)

    def lane_shape(self, lane: int) -> np.ndarray:
        a, b = self.lane_shape_off[lane], self.lane_shape_off[lane + 1]
        return np.stack([self.shape_x[a:b], self.shape_y[a:b]], axis=1)

    def table_bytes(self) -> int:
        return int(sum(v.nbytes for v in self.__dict__.values() if isinstance(v, np.ndarray)))


def _build_lanepoints(net: SumoNet, lane_no: Dict[str, int], out_lanes: List[List[int]], spacing: float):
    """Lanepoint generation in array form.  Returns columns + CSR adjacency."""
    lanes = net.all_lanes()
    # ---------- pass 1: shape lanepoints, BFS over lane connectivity ----------
    s_x: List[float] = []
    s_y: List[float] = []
    s_ang: List[float] = []  # angle handed to the quaternion (already wrapped for shape points)
    s_lane: List[int] = []
    s_next: List[List[int]] = []
    first_of_lane: Dict[int, int] = {}

    def new_shape(x, y, ang, lane):
        s_x.append(x)
        s_y.append(y)
        s_ang.append(ang)
        s_lane.append(lane)
        s_next.append([])
        return len(s_x) - 1

    order: List[int] = []
    for edge in net.getEdges(False):
        for start in edge.getLanes():
            q = deque([(lane_no[start.getID()], -1)])
            while q:
                li, prev = q.popleft()
                if li in first_of_lane:
                    if prev >= 0:
                        s_next[prev].append(first_of_lane[li])
                    continue
                shape = lanes[li].getShape(False)
                n = len(shape)
                assert n >= 2
                ang0 = _wrap(_vec_to_radians(shape[1][0] - shape[0][0], shape[1][1] - shape[0][1]))
                cur = new_shape(shape[0][0], shape[0][1], ang0, li)
                first_of_lane[li] = cur
                order.append(cur)
                if prev >= 0:
                    s_next[prev].append(cur)
                for k in range(1, n - 1):
                    ang = _wrap(_vec_to_radians(shape[k + 1][0] - shape[k][0], shape[k + 1][1] - shape[k][1]))
                    nxt = new_shape(shape[k][0], shape[k][1], ang, li)
                    order.append(nxt)
                    s_next[cur].append(nxt)
                    cur = nxt
                last = new_shape(shape[-1][0], shape[-1][1], s_ang[cur], li)
                order.append(last)
                s_next[cur].append(last)
                cur = last
                for ol in _bfs_out(lanes[li], net, lane_no):
                    q.append((ol, cur))

    # ---------- pass 2: interpolation at `spacing`, BFS over the shape graph ----------
    x: List[float] = []
    y: List[float] = []
    ang: List[float] = []
    lane_col: List[int] = []
    inferred: List[int] = []
    nexts: List[List[int]] = []
    memo: Dict[tuple, int] = {}

    def new_lp(px, py, a, lane, inf):
        x.append(px)
        y.append(py)
        ang.append(a)
        lane_col.append(lane)
        inferred.append(inf)
        nexts.append([])
        return len(x) - 1

    def key(s):
        half = s_ang[s] * 0.5
        return (s_lane[s], s_x[s], s_y[s], math.sin(half), math.cos(half))

    last_threshold = 0.8 * spacing
    min_dist_next_shape = 1.4
    for root in order:
        q = deque([(root, -1)])
        while q:
            s, prev = q.popleft()
            k = key(s)
            first = memo.get(k)
            if first is not None:
                if prev >= 0:
                    nexts[prev].append(first)
                continue
            first = new_lp(s_x[s], s_y[s], s_ang[s], s_lane[s], 0)
            if prev >= 0:
                nexts[prev].append(first)
            memo[k] = first
            for t in s_next[s]:
                if s_lane[t] == s_lane[s] or s_lane[t] in out_lanes[s_lane[s]]:
                    # walk the segment s -> t
                    cur = first
                    vx, vy = s_x[t] - s_x[s], s_y[t] - s_y[s]
                    seg_len = float(np.linalg.norm(np.array([vx, vy])))
                    dist = spacing
                    while dist < seg_len:
                        p = dist / seg_len
                        px, py = s_x[s] + vx * p, s_y[s] + vy * p
                        half = float(np.linalg.norm(0.5 * (np.array([x[cur], y[cur]]) - np.array([s_x[t], s_y[t]]))))
                        if half < min_dist_next_shape:
                            mid = 0.5 * (np.array([s_x[t], s_y[t]]) + np.array([x[cur], y[cur]]))
                            px, py = float(mid[0]), float(mid[1])
                        dn = float(np.linalg.norm(np.array([s_x[t], s_y[t]]) - np.array([px, py])))
                        if dn < last_threshold:
                            break
                        nl = new_lp(px, py, _vec_to_radians(vx, vy), s_lane[s], 1)
                        nexts[cur].append(nl)
                        cur = nl
                        dist += spacing
                    q.append((t, cur))
                else:
                    q.append((t, first))
    return x, y, ang, lane_col, inferred, nexts


def _bfs_out(lane, net, lane_no):
    out = []
    for conn in lane.getOutgoing():
        via = conn.getViaLaneID()
        tgt = net.getLane(via) if via else conn.getToLane()
        out.append(lane_no[tgt.getID()])
    return out


def _grid(points_bbox, cell, pad):
    (x0, y0, x1, y1) = points_bbox
    ox, oy = math.floor(x0 - pad), math.floor(y0 - pad)
    nx = int(math.ceil((x1 + pad - ox) / cell)) + 1
    ny = int(math.ceil((y1 + pad - oy) / cell)) + 1
    return float(ox), float(oy), nx, ny


def _csr(lists, n_cells):
    off = np.zeros(n_cells + 1, dtype=np.int32)
    for c, items in lists.items():
        off[c + 1] = len(items)
    off = np.cumsum(off, dtype=np.int64).astype(np.int32)
    idx = np.zeros(int(off[-1]), dtype=np.int32)
    for c, items in lists.items():
        idx[off[c] : off[c] + len(items)] = items
    return off, idx


def compile_map(net: SumoNet, lanepoint_spacing: float = 1.0, default_lane_width: float = 3.2,
                lp_cell: float = 4.0, seg_cell: float = 8.0) -> CompiledMap:
    import os

    lp_cell = float(os.environ.get("SMX_DEV_LP_CELL", lp_cell))  # developer experiments only
    seg_cell = float(os.environ.get("SMX_DEV_SEG_CELL", seg_cell))
    lanes = net.all_lanes()
    lane_no = {l.getID(): i for i, l in enumerate(lanes)}
    edges = net.getEdges(True)
    road_no = {e.getID(): i for i, e in enumerate(edges)}

    out_lanes = [_bfs_out(l, net, lane_no) for l in lanes]
    lane_out_off = np.zeros(len(lanes) + 1, dtype=np.int32)
    lane_out_off[1:] = np.cumsum([len(o) for o in out_lanes])
    lane_out_idx = np.array([j for o in out_lanes for j in o], dtype=np.int32)

    in_lanes = [[lane_no[i.getID()] for i in l.getIncoming()] for l in lanes]
    lane_in_off = np.zeros(len(lanes) + 1, dtype=np.int32)
    lane_in_off[1:] = np.cumsum([len(i) for i in in_lanes])
    lane_in_idx = np.array([j for i in in_lanes for j in i], dtype=np.int32)
    # parallel roads: the other edges between the same two junctions (junction-internal edges carry no
    # from / to node: sumolib leaves them None and Road.parallel_roads cannot be asked there — none here)
    par = []
    for e in edges:
        fn, tn = e.getFromNode(), e.getToNode()
        if fn is None or tn is None:
            par.append([])
            continue
        par.append([road_no[o.getID()] for o in fn.getOutgoing()
                    if o.getID() != e.getID() and o.getToNode() is not None and o.getToNode().getID() == tn.getID()])
    road_par_off = np.zeros(len(edges) + 1, dtype=np.int32)
    road_par_off[1:] = np.cumsum([len(q) for q in par])
    road_par_idx = np.array([j for q in par for j in q], dtype=np.int32)

    shape_off = np.zeros(len(lanes) + 1, dtype=np.int32)
    shape_off[1:] = np.cumsum([len(l.shape) for l in lanes])
    shape_x = np.array([p[0] for l in lanes for p in l.shape], dtype=np.float64)
    shape_y = np.array([p[1] for l in lanes for p in l.shape], dtype=np.float64)

    road_lane_off = np.zeros(len(edges) + 1, dtype=np.int32)
    road_lane_off[1:] = np.cumsum([len(e.lanes) for e in edges])
    road_lanes = np.array([lane_no[l.getID()] for e in edges for l in e.lanes], dtype=np.int32)
    road_out = np.full(len(edges), -1, dtype=np.int32)
    for i, e in enumerate(edges):
        outs = list(e.getOutgoing().keys())
        if outs:
            road_out[i] = road_no[outs[0].getID()]

    x, y, ang, lane_col, inferred, nexts = _build_lanepoints(net, lane_no, out_lanes, lanepoint_spacing)
    n_lp = len(x)
    lp_heading = np.array([_pose_heading(a) for a in ang], dtype=np.float64)
    # unit direction of the heading, radians_to_vec (math.py:247-253)
    dirang = [(h + math.pi * 0.5) % TWO_PI for h in lp_heading]
    lp_dirx = np.array([math.cos(a) for a in dirang], dtype=np.float64)
    lp_diry = np.array([math.sin(a) for a in dirang], dtype=np.float64)
    next_off = np.zeros(n_lp + 1, dtype=np.int32)
    next_off[1:] = np.cumsum([len(n) for n in nexts])
    next_idx = np.array([j for n in nexts for j in n], dtype=np.int32)
    lp_x = np.array(x, dtype=np.float64)
    lp_y = np.array(y, dtype=np.float64)

    # ---- lanepoint grid ----
    bbox = (float(min(shape_x.min(), lp_x.min())), float(min(shape_y.min(), lp_y.min())),
            float(max(shape_x.max(), lp_x.max())), float(max(shape_y.max(), lp_y.max())))
    gx0, gy0, gnx, gny = _grid(bbox, lp_cell, lp_cell)
    cells: Dict[int, List[int]] = {}
    cx = np.floor((lp_x - gx0) / lp_cell).astype(np.int64)
    cy = np.floor((lp_y - gy0) / lp_cell).astype(np.int64)
    for i in range(n_lp):
        cells.setdefault(int(cy[i] * gnx + cx[i]), []).append(i)
    lpg_off, lpg_idx = _csr(cells, gnx * gny)

    # ---- segment grid ----
    seg_lane = []
    seg_v0 = []
    for li, l in enumerate(lanes):
        for k in range(len(l.shape) - 1):
            seg_lane.append(li)
            seg_v0.append(int(shape_off[li]) + k)
    seg_lane = np.array(seg_lane, dtype=np.int32)
    seg_v0 = np.array(seg_v0, dtype=np.int32)
    sx0, sy0, snx, sny = _grid(bbox, seg_cell, seg_cell)
    scells: Dict[int, List[int]] = {}
    for s in range(len(seg_lane)):
        a = seg_v0[s]
        xa, xb = sorted((shape_x[a], shape_x[a + 1]))
        ya, yb = sorted((shape_y[a], shape_y[a + 1]))
        c0x, c1x = int(math.floor((xa - sx0) / seg_cell)), int(math.floor((xb - sx0) / seg_cell))
        c0y, c1y = int(math.floor((ya - sy0) / seg_cell)), int(math.floor((yb - sy0) / seg_cell))
        for iy in range(c0y, c1y + 1):
            for ix in range(c0x, c1x + 1):
                scells.setdefault(iy * snx + ix, []).append(s)
    sg_off, sg_idx = _csr(scells, snx * sny)

    return CompiledMap(
        lane_ids=[l.getID() for l in lanes],
        lane_road=np.array([road_no[l.getEdge().getID()] for l in lanes], dtype=np.int32),
        lane_index=np.array([l.getIndex() for l in lanes], dtype=np.int32),
        lane_width=np.array([l.getWidth() for l in lanes], dtype=np.float64),
        lane_speed=np.array([l.getSpeed() for l in lanes], dtype=np.float64),
        lane_length=np.array([l.getLength() for l in lanes], dtype=np.float64),
        lane_in_junction=np.array([1 if l.getEdge().isSpecial() else 0 for l in lanes], dtype=np.uint8),
        lane_shape_off=shape_off,
        shape_x=shape_x,
        shape_y=shape_y,
        lane_out_off=lane_out_off,
        lane_out_idx=lane_out_idx,
        lane_in_off=lane_in_off,
        lane_in_idx=lane_in_idx,
        road_par_off=road_par_off,
        road_par_idx=road_par_idx,
        road_ids=[e.getID() for e in edges],
        road_lane_off=road_lane_off,
        road_lanes=road_lanes,
        road_is_junction=np.array([1 if e.isSpecial() else 0 for e in edges], dtype=np.uint8),
        road_out_road=road_out,
        lp_x=lp_x,
        lp_y=lp_y,
        lp_heading=lp_heading,
        lp_dirx=lp_dirx,
        lp_diry=lp_diry,
        lp_lane=np.array(lane_col, dtype=np.int32),
        lp_inferred=np.array(inferred, dtype=np.uint8),
        lp_next_off=next_off,
        lp_next_idx=next_idx,
        lpg_origin=np.array([gx0, gy0], dtype=np.float64),
        lpg_cell=float(lp_cell),
        lpg_dims=np.array([gnx, gny], dtype=np.int32),
        lpg_off=lpg_off,
        lpg_idx=lpg_idx,
        seg_lane=seg_lane,
        seg_v0=seg_v0,
        sg_origin=np.array([sx0, sy0], dtype=np.float64),
        sg_cell=float(seg_cell),
        sg_dims=np.array([snx, sny], dtype=np.int32),
        sg_off=sg_off,
        sg_idx=sg_idx,
        default_lane_width=default_lane_width,
        lanepoint_spacing=lanepoint_spacing,
        max_fanout=int(max((len(n) for n in nexts), default=1)),
        shifted_by=tuple(net.shifted_by),
    )


# --------------------------------------------------------------------------------------
# packed device records (what smx_load_map uploads)
# --------------------------------------------------------------------------------------
LP_REC = np.dtype(
    [("x", "<f8"), ("y", "<f8"), ("heading", "<f8"), ("dirx", "<f8"), ("diry", "<f8"), ("lane", "<i4"),
     ("next_off", "<i4"), ("next0", "<i4"), ("n_next", "<u2"), ("inferred", "u1"), ("flags", "u1"),
     ("knot_next", "<i4"), ("knot_hops", "<i4")], align=False)
SHAPE_REC = np.dtype([("x", "<f8"), ("y", "<f8"), ("cum", "<f8"), ("len", "<f8")], align=False)
SUCC_REC = np.dtype([("idx", "<i4"), ("lane", "<i4"), ("knot", "<i4"), ("hops", "<i4")], align=False)
PT_REC = np.dtype([("x", "<f8"), ("y", "<f8"), ("idx", "<i4"), ("lane", "<i4")], align=False)
SEG_REC = np.dtype([("x1", "<f8"), ("y1", "<f8"), ("x2", "<f8"), ("y2", "<f8"), ("thr", "<f8"), ("len", "<f8"), ("cum", "<f8"),
                    ("lane", "<i4"), ("v0", "<i4")], align=False)
assert LP_REC.itemsize == 64 and SUCC_REC.itemsize == 16 and PT_REC.itemsize == 24 and SEG_REC.itemsize == 64


def pack_tables(cm: CompiledMap) -> Dict[str, np.ndarray]:
    """Array-of-records forms of the lanepoint graph and the two grids.

    * ``lp_rec``: one 64-byte record per lanepoint = one cache line per hop of a path walk,
      with *knot skip-links*: ``knot_next`` / ``knot_hops`` give the next non-inferred
      lanepoint down the (single-successor) chain, so a 32-hop walk touches only the shape
      points it passes.  Interpolated lanepoints of one segment have consecutive indices
      (``lanepoints.py:462-515`` appends them in one loop), which lets the walk address the
      point it stops on directly; ``flags & 1`` certifies that for the record's chain.
    * ``succ_rec``: per successor of a branching lanepoint: first point, its lane, and the
      knot that branch reaches.
    * ``lpg_pts`` / ``sg_rec``: the grid cells' members stored by value, contiguous per cell,
      so a cell scan is one stream instead of an index chase.
    """
    n = cm.n_lanepoints
    off, idx = cm.lp_next_off, cm.lp_next_idx
    inferred = cm.lp_inferred.astype(bool)
    rec = np.zeros(n, dtype=LP_REC)
    rec["x"], rec["y"], rec["heading"] = cm.lp_x, cm.lp_y, cm.lp_heading
    rec["dirx"], rec["diry"], rec["lane"] = cm.lp_dirx, cm.lp_diry, cm.lp_lane
    rec["next_off"] = off[:-1]
    rec["n_next"] = np.diff(off)
    rec["inferred"] = cm.lp_inferred
    rec["next0"] = -1
    rec["knot_next"] = -1
    rec["knot_hops"] = 0

    def chase(first):
        """From `first`, follow single successors until a non-inferred point: (knot, hops, consecutive)."""
        hops, cur, consecutive = 1, first, True
        while inferred[cur]:
            assert off[cur + 1] - off[cur] == 1, "an interpolated lanepoint must have exactly one successor"
            nxt = int(idx[off[cur]])
            if inferred[nxt] and nxt != cur + 1:
                consecutive = False
            cur = nxt
            hops += 1
        return cur, hops, consecutive

    succ = np.zeros(len(idx), dtype=SUCC_REC)
    for i in range(n):
        a, b = int(off[i]), int(off[i + 1])
        if b - a >= 1:
            rec["next0"][i] = idx[a]
        for k in range(a, b):
            first = int(idx[k])
            knot, hops, ok = chase(first)
            succ[k] = (first, cm.lp_lane[first], knot, hops)
            if k == a:
                rec["knot_next"][i], rec["knot_hops"][i] = knot, hops
                rec["flags"][i] = 1 if ok else 0
            elif not ok:
                rec["flags"][i] = 0
    # grid cells by value
    pts = np.zeros(len(cm.lpg_idx), dtype=PT_REC)
    pts["idx"] = cm.lpg_idx
    pts["x"], pts["y"], pts["lane"] = cm.lp_x[cm.lpg_idx], cm.lp_y[cm.lpg_idx], cm.lp_lane[cm.lpg_idx]
    seg = np.zeros(len(cm.sg_idx), dtype=SEG_REC)
    v0 = cm.seg_v0[cm.sg_idx]
    lane = cm.seg_lane[cm.sg_idx]
    seg["x1"], seg["y1"], seg["x2"], seg["y2"] = cm.shape_x[v0], cm.shape_y[v0], cm.shape_x[v0 + 1], cm.shape_y[v0 + 1]
    # road_with_point threshold, the reference's expression (sumo_road_network.py:707)
    seg["thr"] = np.array([0.5 * float(w) + 1e-1 for w in cm.lane_width[lane]])
    seg["lane"] = lane
    seg["v0"] = v0
    # centre-line vertices with their running arclength: the sums follow the reference's loops
    # (utils/math.py:319-331, 370-390: `seen += length` vertex by vertex), so offsets compare equal
    shp = np.zeros(len(cm.shape_x), dtype=SHAPE_REC)
    shp["x"], shp["y"] = cm.shape_x, cm.shape_y
    for lane in range(cm.n_lanes):
        a, b = int(cm.lane_shape_off[lane]), int(cm.lane_shape_off[lane + 1])
        acc = 0.0
        for v in range(a, b):
            shp["cum"][v] = acc
            if v + 1 < b:
                ex, ey = float(cm.shape_x[v] - cm.shape_x[v + 1]), float(cm.shape_y[v] - cm.shape_y[v + 1])
                d = math.sqrt(ex * ex + ey * ey)
                shp["len"][v] = d
                acc = acc + d
    seg["len"], seg["cum"] = shp["len"][v0], shp["cum"][v0]  # (one cache line per segment: no vertex look-up behind it)
    return dict(lp_rec=rec, succ_rec=succ, lpg_pts=pts, sg_rec=seg, shape_rec=shp)


_MAP_ARRAYS = [
    ("lane_road", np.int32), ("lane_index", np.int32), ("lane_width", np.float64), ("lane_speed", np.float64),
    ("lane_length", np.float64), ("lane_in_junction", np.uint8), ("lane_shape_off", np.int32),
    ("shape_x", np.float64), ("shape_y", np.float64), ("lane_out_off", np.int32), ("lane_out_idx", np.int32),
    ("road_lane_off", np.int32), ("road_lanes", np.int32), ("road_is_junction", np.uint8),
    ("road_out_road", np.int32), ("lpg_off", np.int32), ("sg_off", np.int32),
    ("lane_in_off", np.int32), ("lane_in_idx", np.int32), ("road_par_off", np.int32), ("road_par_idx", np.int32),
]


def map_tables_struct(cm: "CompiledMap"):
    """Fill ``smx_map_tables`` with pointers into (kept-alive) contiguous numpy arrays."""
    from ._native import SmxMapTables

    keep = []
    t = SmxMapTables()
    packed = cm.extras.get("packed") or pack_tables(cm)  # scenario_build.load_compiled_map attaches them
    t.n_lanes, t.n_roads = cm.n_lanes, len(cm.road_ids)
    t.n_lanepoints, t.n_shape_pts, t.n_succ = cm.n_lanepoints, len(cm.shape_x), len(packed["succ_rec"])
    for name, dt in _MAP_ARRAYS:
        arr = np.ascontiguousarray(getattr(cm, name), dtype=dt)
        if arr.size == 0:
            arr = np.zeros(1, dtype=dt)
        keep.append(arr)
        setattr(t, name, arr.ctypes.data)
    for name in ("lp_rec", "succ_rec", "lpg_pts", "sg_rec", "shape_rec"):
        arr = np.ascontiguousarray(packed[name])
        if arr.size == 0:
            arr = np.zeros(1, dtype=arr.dtype)
        keep.append(arr)
        setattr(t, name, arr.ctypes.data)
    t.lpg_x0, t.lpg_y0, t.lpg_cell = float(cm.lpg_origin[0]), float(cm.lpg_origin[1]), float(cm.lpg_cell)
    t.lpg_nx, t.lpg_ny = int(cm.lpg_dims[0]), int(cm.lpg_dims[1])
    t.sg_x0, t.sg_y0, t.sg_cell = float(cm.sg_origin[0]), float(cm.sg_origin[1]), float(cm.sg_cell)
    t.sg_nx, t.sg_ny = int(cm.sg_dims[0]), int(cm.sg_dims[1])
    t.default_lane_width = float(cm.default_lane_width)
    return t, keep
