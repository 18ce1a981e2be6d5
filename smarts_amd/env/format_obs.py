"""``FormatObs`` / ``StdObs`` mirror (reference ``smarts/env/wrappers/format_obs.py:40-603``): fixed
shape numpy observations.  The device already writes this layout (include/smx.h), so
``FormatObs.from_rows`` slices the dense rows directly; ``FormatObs.observation`` converts
``Observation`` objects with the reference's padding rules for code that holds objects."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, Optional, Union

import numpy as np

from .. import _native as nat
from .custom_observations import lane_ttc
from .observations import Observation

_LIDAR_SHP = 300
_NEIGHBOR_SHP = 10
_WAYPOINT_SHP = (4, 20)


@dataclass(frozen=True)
class StdObs:
    """format_obs.py:45-197: same fields, dtypes and defaults."""

    dist: np.float32
    ego: Dict[str, Union[np.int8, np.float32, np.ndarray]]
    events: Dict[str, np.int8]
    dagm: Optional[np.ndarray] = None
    lidar: Optional[Dict[str, np.ndarray]] = None
    neighbors: Optional[Dict[str, np.ndarray]] = None
    ogm: Optional[np.ndarray] = None
    rgb: Optional[np.ndarray] = None
    ttc: Optional[Dict[str, Union[np.float32, np.ndarray]]] = None
    waypoints: Optional[Dict[str, np.ndarray]] = None


def _pad(a: np.ndarray, shape) -> np.ndarray:
    out = np.zeros(shape, dtype=a.dtype)
    sl = tuple(slice(0, min(s, t)) for s, t in zip(a.shape, shape))
    out[sl] = a[sl]
    return out


def _std_ego(v) -> Dict[str, Any]:
    """format_obs.py:401-435."""
    z = np.zeros((3,), dtype=np.float32)
    f32 = lambda a: z.copy() if a is None else np.asarray(a).astype(np.float32)  # noqa: E731
    return {
        "angular_acceleration": f32(v.angular_acceleration), "angular_jerk": f32(v.angular_jerk),
        "angular_velocity": np.asarray(v.angular_velocity).astype(np.float32),
        "box": np.array(v.bounding_box.as_lwh).astype(np.float32), "heading": np.float32(v.heading),
        "lane_index": np.int8(v.lane_index if v.lane_index is not None else 0),
        "linear_acceleration": f32(v.linear_acceleration), "linear_jerk": f32(v.linear_jerk),
        "linear_velocity": np.asarray(v.linear_velocity).astype(np.float32),
        "pos": np.asarray(v.position).astype(np.float64), "speed": np.float32(v.speed),
        "steering": np.float32(v.steering), "yaw_rate": np.float32(v.yaw_rate),
    }


def _std_events(ev) -> Dict[str, np.int8]:
    """format_obs.py:438-449."""
    return {
        "agents_alive_done": np.int8(ev.agents_alive_done), "collisions": np.int8(len(ev.collisions) > 0),
        "not_moving": np.int8(ev.not_moving), "off_road": np.int8(ev.off_road), "off_route": np.int8(ev.off_route),
        "on_shoulder": np.int8(ev.on_shoulder), "reached_goal": np.int8(ev.reached_goal),
        "reached_max_episode_steps": np.int8(ev.reached_max_episode_steps), "wrong_way": np.int8(ev.wrong_way),
    }


def _std_lidar(val) -> Optional[Dict[str, np.ndarray]]:
    """format_obs.py:452-489 (misses become zeros)."""
    if not val:
        return None
    hit = np.array(val[1], dtype=np.int8)
    cloud = np.nan_to_num(np.array(val[0], dtype=np.float64), copy=False, nan=0.0, posinf=0.0, neginf=0.0)
    origin, vector = zip(*val[2])
    origin, vector = np.array(origin, np.float64), np.array(vector, np.float64)
    if not (hit.shape == (_LIDAR_SHP,) and cloud.shape == origin.shape == vector.shape == (_LIDAR_SHP, 3)):
        raise Exception("Internal Error: Mismatched lidar point cloud shape.")
    return {"hit": hit, "point_cloud": cloud, "ray_origin": origin, "ray_vector": vector}


def _std_neighbors(nghbs) -> Optional[Dict[str, np.ndarray]]:
    """format_obs.py:492-536: the FIRST ten, zero padded."""
    if not nghbs:
        return None
    nghbs = nghbs[:_NEIGHBOR_SHP]
    n = _NEIGHBOR_SHP
    return {
        "box": _pad(np.array([v.bounding_box.as_lwh for v in nghbs], dtype=np.float32), (n, 3)),
        "heading": _pad(np.array([v.heading for v in nghbs], dtype=np.float32), (n,)),
        "lane_index": _pad(np.array([v.lane_index if v.lane_index is not None else 0 for v in nghbs], dtype=np.int8), (n,)),
        "pos": _pad(np.array([v.position for v in nghbs], dtype=np.float64), (n, 3)),
        "speed": _pad(np.array([v.speed for v in nghbs], dtype=np.float32), (n,)),
    }


def _std_waypoints(paths) -> Optional[Dict[str, np.ndarray]]:
    """format_obs.py:565-603: first 4 paths x first 20 waypoints, zero padded; pos gets a zero z."""
    if not paths:
        return None
    P, W = _WAYPOINT_SHP
    out = {
        "heading": np.zeros((P, W), np.float32), "lane_index": np.zeros((P, W), np.int8),
        "lane_width": np.zeros((P, W), np.float32), "pos": np.zeros((P, W, 3), np.float64),
        "speed_limit": np.zeros((P, W), np.float32),
    }
    n_first = len(paths[0])
    for p, path in enumerate(paths[:P]):
        # np.array over ragged paths fails in the reference; paths of one query share their length
        for w, wp in enumerate(path[:min(W, n_first)]):
            out["heading"][p, w] = wp.heading
            out["lane_index"][p, w] = wp.lane_index
            out["lane_width"][p, w] = wp.lane_width
            out["pos"][p, w, :2] = wp.pos
            out["speed_limit"][p, w] = wp.speed_limit
    return out


def _std_ttc(obs: Observation) -> Optional[Dict[str, Any]]:
    """format_obs.py:551-562."""
    if not obs.neighborhood_vehicle_states or not obs.waypoint_paths:
        return None
    val = lane_ttc(obs)
    return {
        "angle_error": np.float32(val["angle_error"][0]),
        "distance_from_center": np.float32(val["distance_from_center"][0]),
        "dtc": np.array(val["ego_lane_dist"], dtype=np.float32),
        "ttc": np.array(val["ego_ttc"], dtype=np.float32),
    }


def std_obs(obs: Observation) -> StdObs:
    """One ``Observation`` -> ``StdObs`` (format_obs.py:261-282)."""
    return StdObs(
        dist=np.float32(obs.distance_travelled), ego=_std_ego(obs.ego_vehicle_state), events=_std_events(obs.events),
        dagm=obs.drivable_area_grid_map.data.astype(np.uint8) if obs.drivable_area_grid_map else None,
        lidar=_std_lidar(obs.lidar_point_cloud), neighbors=_std_neighbors(obs.neighborhood_vehicle_states),
        ogm=obs.occupancy_grid_map.data.astype(np.uint8) if obs.occupancy_grid_map else None, rgb=None,
        ttc=_std_ttc(obs), waypoints=_std_waypoints(obs.waypoint_paths),
    )


class FormatObs:
    """Environment wrapper: observations become ``Dict[agent_id, StdObs]`` (format_obs.py:200-282).
    As in the reference, every agent must share one ``AgentInterface`` and observation adapters
    must not be used inside the wrapped env."""

    def __init__(self, env):
        self.env = env
        specs = env.agent_specs
        first = next(iter(specs.values())).interface
        for name in ("accelerometer", "drivable_area_grid_map", "lidar", "neighborhood_vehicles", "ogm", "rgb",
                     "waypoints"):
            val = getattr(first, name)
            assert all(getattr(s.interface, name) == val for s in specs.values()), \
                f"To use FormatObs wrapper, all agents must have the same AgentInterface.{name} attribute."

    def __getattr__(self, name):
        return getattr(self.env, name)

    def observation(self, obs: Dict[str, Observation]) -> Dict[str, StdObs]:
        return {agent_id: std_obs(o) for agent_id, o in obs.items()}

    def reset(self):
        return self.observation(self.env.reset())

    def step(self, actions):
        obs, rewards, dones, infos = self.env.step(actions)
        return self.observation(obs), rewards, dones, infos

    def close(self):
        return self.env.close()

    # -------------------------------------------------------------- dense rows -> StdObs, no objects
    @staticmethod
    def from_rows(rows: Dict[str, np.ndarray], env: int, slot: int) -> StdObs:
        """Slice ``StdObs`` of agent (env, slot) straight out of host copies of the dense device
        rows (the ttc block needs lane ids per neighbour and is only built by ``observation``)."""
        E = nat.EGO
        f = rows["ego_f32"][env, slot]
        v3 = lambda k: np.array(f[E[k]:E[k] + 3], dtype=np.float32)  # noqa: E731
        ego = {
            "angular_acceleration": v3("ANG_ACC"), "angular_jerk": v3("ANG_JERK"), "angular_velocity": v3("ANG_VEL"),
            "box": v3("BOX"), "heading": np.float32(f[E["HEADING"]]),
            "lane_index": np.int8(max(int(rows["ego_lane"][env, slot, 1]), 0)),
            "linear_acceleration": v3("LIN_ACC"), "linear_jerk": v3("LIN_JERK"), "linear_velocity": v3("LIN_VEL"),
            "pos": np.array(rows["ego_pos"][env, slot], dtype=np.float64), "speed": np.float32(f[E["SPEED"]]),
            "steering": np.float32(f[E["STEERING"]]), "yaw_rate": np.float32(f[E["YAW_RATE"]]),
        }
        ev = rows["events"][env, slot]
        events = {name: np.int8(ev[i]) for i, name in enumerate(nat.EVENT_NAMES)}
        neighbors = waypoints = ogm = lidar = None
        if "nb_pos" in rows and rows["nb_count"][env, slot] > 0:
            neighbors = {
                "box": np.array(rows["nb_box"][env, slot]), "heading": np.array(rows["nb_heading"][env, slot]),
                "lane_index": np.maximum(rows["nb_lane_index"][env, slot], 0).astype(np.int8),
                "pos": np.array(rows["nb_pos"][env, slot]), "speed": np.array(rows["nb_speed"][env, slot]),
            }
        if "wp_pos" in rows and rows["wp_count"][env, slot, 0] > 0:
            # rows of any window -> the fixed StdObs (4, 20) block (format_obs.py:565-603)
            def window(key):
                a = rows[key][env, slot]
                return _pad(np.array(a[:_WAYPOINT_SHP[0], :_WAYPOINT_SHP[1]]), _WAYPOINT_SHP + a.shape[2:])

            waypoints = {
                "heading": window("wp_heading"), "lane_index": window("wp_lane_index"),
                "lane_width": window("wp_lane_width"), "pos": window("wp_pos"), "speed_limit": window("wp_speed_limit"),
            }
        if "ogm" in rows:
            ogm = np.array(rows["ogm"][env, slot], dtype=np.uint8)[..., None]
        dagm = np.array(rows["dagm"][env, slot], dtype=np.uint8)[..., None] if "dagm" in rows else None
        if "lidar_hit" in rows:
            hit = rows["lidar_hit"][env, slot].astype(np.int8)
            cloud = np.nan_to_num(np.array(rows["lidar_point"][env, slot]), nan=0.0, posinf=0.0, neginf=0.0)
            lidar = {"hit": hit, "point_cloud": cloud}
        return StdObs(dist=np.float32(rows["dist"][env, slot]), ego=ego, events=events, lidar=lidar, neighbors=neighbors,
                      ogm=ogm, waypoints=waypoints, dagm=dagm)
