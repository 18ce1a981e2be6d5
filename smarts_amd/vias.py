"""Via points (host side): the scenario-studio ``Via`` type and its resolution against a compiled map.

Mirrors ``smarts/sstudio/types.py:422-435`` (what a scenario author writes) and
``Scenario.to_scenario_via`` (``smarts/core/scenario.py:652-676``: road + lane index + offset ->
lane id, position on the centre line, hit distance defaulting to half the lane width), producing the
records the device's via sensor reads (``include/smx.h`` ``smx_via``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np

from .map_compiler import CompiledMap


@dataclass(frozen=True)
class Via:
    """sstudio/types.py:422-435."""

    road_id: str
    lane_index: int
    lane_offset: float
    required_speed: float
    hit_distance: float = -1  # negative: half the lane width


@dataclass(frozen=True)
class ResolvedVia:
    """plan.py:180-188 (``Via``) with the lane as a table index."""

    lane: int
    lane_id: str
    road_id: str
    lane_index: int
    position: Tuple[float, float]
    hit_distance: float
    required_speed: float


def _position_at_shape_offset(shape: np.ndarray, offset: float) -> Tuple[float, float]:
    """utils/math.py:300-331 (position_at_offset inside position_at_shape_offset), restated."""

    def isclose(a, b):
        return abs(a - b) <= max(1e-09 * max(abs(a), abs(b)), 0.0)

    seen = 0.0
    cur = shape[0]
    for nxt in shape[1:]:
        ex, ey = float(cur[0] - nxt[0]), float(cur[1] - nxt[1])
        length = math.sqrt(ex * ex + ey * ey)
        if seen + length > offset:
            local = offset - seen
            if isclose(local, 0.0):
                return float(cur[0]), float(cur[1])
            if isclose(length, local):
                return float(nxt[0]), float(nxt[1])
            return (float(cur[0] + (nxt[0] - cur[0]) * (local / length)),
                    float(cur[1] + (nxt[1] - cur[1]) * (local / length)))
        seen += length
        cur = nxt
    return float(shape[-1][0]), float(shape[-1][1])


def resolve_vias(cm: CompiledMap, vias: Sequence[Via]) -> List[ResolvedVia]:
    out = []
    for via in vias:
        if via.road_id not in cm.road_ids:
            raise ValueError(f"via: unknown road {via.road_id!r}")
        road = cm.road_ids.index(via.road_id)
        lanes = cm.road_lanes[cm.road_lane_off[road]:cm.road_lane_off[road + 1]]
        if not 0 <= via.lane_index < len(lanes):
            raise ValueError(f"via: road {via.road_id!r} has no lane {via.lane_index}")
        lane = int(lanes[via.lane_index])
        width = float(cm.lane_width[lane])
        hit = via.hit_distance if via.hit_distance > 0 else width / 2  # scenario.py:659-662
        pos = _position_at_shape_offset(cm.lane_shape(lane), float(via.lane_offset))
        out.append(ResolvedVia(lane=lane, lane_id=cm.lane_ids[lane], road_id=via.road_id, lane_index=via.lane_index,
                               position=pos, hit_distance=float(hit), required_speed=float(via.required_speed)))
    return out
