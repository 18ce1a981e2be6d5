"""``lane_ttc`` (reference ``smarts/env/custom_observations.py:148-280``): time / distance to
collision on the right, current and left lane from an agent's ``Observation``.  Host-side; consumed
by ``FormatObs`` for the ``ttc`` block of ``StdObs``."""
from __future__ import annotations

from typing import Dict

import numpy as np

from .observations import Observation


def _path_arclengths(path):
    """Cumulative distance of each waypoint from the first one of its path (:205-212)."""
    out, acc = [], 0.0
    for a, b in zip(path, path[1:]):
        out.append(acc)
        acc += np.linalg.norm(b.pos - a.pos)
    out.append(acc)
    return out


def _ttc_by_path(obs: Observation):
    """:195-254.  For every neighbour: the waypoint on the neighbour's lane nearest to it (over all
    paths); if within 2 m, the arclength to it is the gap.  Units follow the reference: relative
    speed is scaled by 1000/3600, ttc by 1/10, gap by 1/100; defaults 1000 and 1."""
    ego, paths = obs.ego_vehicle_state, obs.waypoint_paths
    flat = [(wp, pi, d) for pi, path in enumerate(paths) for wp, d in zip(path, _path_arclengths(path))]
    ttc_by_path = [1000] * len(paths)
    dist_by_path = [1] * len(paths)
    for v in obs.neighborhood_vehicle_states:
        vpos = np.asarray(v.position, dtype=np.float64)[:2]
        best = None
        for wp, pi, d in flat:
            if wp.lane_id != v.lane_id:
                continue
            gap = np.linalg.norm(wp.pos - vpos)
            if best is None or gap < best[0]:  # min() keeps the first of equal keys
                best = (gap, pi, d)
        if best is None or best[0] > 2:
            continue
        _, pi, lane_dist = best
        rel = (ego.speed - v.speed) * 1000 / 3600
        if abs(rel) < 1e-5:
            rel = 1e-5
        ttc = lane_dist / rel / 10
        if ttc <= 0:
            continue
        dist_by_path[pi] = min(dist_by_path[pi], lane_dist / 100)
        ttc_by_path[pi] = min(ttc_by_path[pi], ttc)
    return ttc_by_path, dist_by_path


def _three_lanes(ego_lane_index: int, values):
    """:257-280: [right, current, left], indexed — as the reference does — by *lane index* into the
    per-path list; 0 where there is no such entry."""
    out = [0, values[ego_lane_index], 0]
    if ego_lane_index + 1 <= len(values) - 1:
        out[2] = values[ego_lane_index + 1]
    if ego_lane_index - 1 >= 0:
        out[0] = values[ego_lane_index - 1]
    return out


def lane_ttc(obs: Observation) -> Dict[str, np.ndarray]:
    """:148-184."""
    ego = obs.ego_vehicle_state
    firsts = [path[0] for path in obs.waypoint_paths]
    closest = min(firsts, key=lambda wp: wp.dist_to(ego.position))
    norm_dist = closest.signed_lateral_error(ego.position) / (closest.lane_width * 0.5)
    ttc_by_path, dist_by_path = _ttc_by_path(obs)
    return {
        "distance_from_center": np.array([norm_dist]),
        "angle_error": np.array([closest.relative_heading(ego.heading)]),
        "speed": np.array([ego.speed]),
        "steering": np.array([ego.steering]),
        "ego_ttc": np.array(_three_lanes(closest.lane_index, ttc_by_path)),
        "ego_lane_dist": np.array(_three_lanes(closest.lane_index, dist_by_path)),
    }
