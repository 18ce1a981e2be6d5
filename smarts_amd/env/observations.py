"""Observation types of the reference (``smarts/core/sensors.py:44-203``, ``events.py:23-35``,
``road_map.py:556-618``, ``coordinates.py:46-80, 169-252``) and their construction from the dense
device rows (include/smx.h ``smx_outputs``).  Object construction happens on the host and only when
the object API (``HiWayEnv.step``) is used; the dense tensors are the fast path.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, NamedTuple, Optional, Sequence, Tuple

import numpy as np

from .. import _native as nat


class Heading(float):
    """coordinates.py:169-252: a float in (-pi, pi], 0 = +y, counter-clockwise."""

    def __new__(cls, value=...):
        value = value % (2 * math.pi)
        if value > math.pi:
            value -= 2 * math.pi
        return float.__new__(cls, value)

    def relative_to(self, other: "Heading") -> "Heading":
        """coordinates.py:227-239."""
        return Heading(Heading(self - other))

    def direction_vector(self) -> np.ndarray:
        """coordinates.py:241-243 over radians_to_vec (utils/math.py:247-253)."""
        angle = (self + math.pi * 0.5) % (2 * math.pi)
        return np.array((math.cos(angle), math.sin(angle)))


@dataclass(frozen=True)
class Dimensions:
    """coordinates.py:46-80."""

    length: float
    width: float
    height: float

    @property
    def as_lwh(self) -> Tuple[float, float, float]:
        return (self.length, self.width, self.height)


@dataclass
class Waypoint:
    """road_map.py:556-618."""

    pos: np.ndarray
    heading: Heading
    lane_id: str
    lane_width: float
    speed_limit: float
    lane_index: int

    def __eq__(self, other) -> bool:
        if not isinstance(other, Waypoint):
            return False
        return ((self.pos == other.pos).all() and self.heading == other.heading and self.lane_width == other.lane_width
                and self.speed_limit == other.speed_limit and self.lane_id == other.lane_id
                and self.lane_index == other.lane_index)

    def __hash__(self):
        return hash((*self.pos, self.heading, self.lane_width, self.speed_limit, self.lane_id, self.lane_index))

    def relative_heading(self, h: Heading) -> Heading:
        return self.heading.relative_to(h)

    def signed_lateral_error(self, p) -> float:
        """road_map.py:608-614 over signed_dist_to_line (utils/math.py:163-185): negative right of
        the heading line, positive left."""
        d = self.heading.direction_vector()
        q = np.asarray(p, dtype=np.float64)[:2]
        p1, p2 = self.pos, self.pos + d
        u = abs(d[1] * q[0] - d[0] * q[1] + p2[0] * p1[1] - p2[1] * p1[0])
        dist = u / np.linalg.norm(d)
        return float(dist * np.sign(np.dot(q - p1, np.array([-d[1], d[0]]))))

    def dist_to(self, p) -> float:
        return float(np.linalg.norm(self.pos - np.asarray(p[:2])))


class Events(NamedTuple):
    """events.py:23-35."""

    collisions: Sequence
    off_road: bool
    off_route: bool
    on_shoulder: bool
    wrong_way: bool
    not_moving: bool
    reached_goal: bool
    reached_max_episode_steps: bool
    agents_alive_done: bool


@dataclass
class Collision:
    """sensors.py:206-211: one per vehicle the agent's chassis touched in the last physics update;
    ``collidee_id`` is the actor id of that vehicle's owner (smarts.py:1284-1290).  The device reports the
    collidees of an agent as a slot mask (``smx_outputs.collidees``)."""

    collidee_id: Optional[str]


class VehicleObservation(NamedTuple):
    """sensors.py:44-62."""

    id: str
    position: Tuple[float, float, float]
    bounding_box: Dimensions
    heading: Heading
    speed: float
    road_id: Optional[str]
    lane_id: Optional[str]
    lane_index: Optional[int]


class EgoVehicleObservation(NamedTuple):
    """sensors.py:65-101."""

    id: str
    position: np.ndarray
    bounding_box: Dimensions
    heading: Heading
    speed: float
    steering: float
    yaw_rate: float
    road_id: Optional[str]
    lane_id: Optional[str]
    lane_index: Optional[int]
    mission: object
    linear_velocity: np.ndarray
    angular_velocity: np.ndarray
    linear_acceleration: Optional[np.ndarray]
    angular_acceleration: Optional[np.ndarray]
    linear_jerk: Optional[np.ndarray]
    angular_jerk: Optional[np.ndarray]


class GridMapMetadata(NamedTuple):
    """sensors.py:110-124."""

    created_at: int
    resolution: float
    width: int
    height: int
    camera_pos: Tuple[float, float, float]
    camera_heading_in_degrees: float


class DrivableAreaGridMap(NamedTuple):
    """sensors.py:145-151."""

    metadata: GridMapMetadata
    data: np.ndarray


class OccupancyGridMap(NamedTuple):
    """sensors.py:136-142."""

    metadata: GridMapMetadata
    data: np.ndarray


@dataclass
class ViaPoint:
    """sensors.py:145-157."""

    position: Tuple[float, float]
    lane_index: float
    road_id: str
    required_speed: float


@dataclass(frozen=True)
class Vias:
    """sensors.py:165-172 (missions here carry no vias)."""

    near_via_points: List
    hit_via_points: List


@dataclass
class Observation:
    """sensors.py:178-203."""

    dt: float
    step_count: int
    elapsed_sim_time: float
    events: Events
    ego_vehicle_state: EgoVehicleObservation
    neighborhood_vehicle_states: Optional[List[VehicleObservation]]
    waypoint_paths: Optional[List[List[Waypoint]]]
    distance_travelled: float
    lidar_point_cloud: Optional[Tuple[List[np.ndarray], List[bool], List[Tuple[np.ndarray, np.ndarray]]]]
    drivable_area_grid_map: Optional[object]
    occupancy_grid_map: Optional[OccupancyGridMap]
    top_down_rgb: Optional[object]
    road_waypoints: Optional[object]
    via_data: Vias


class RoadWaypoints(NamedTuple):
    """sensors.py:104-107: per lane id (of the ego road, its parallel roads and the roads oncoming at the vehicle)
    the waypoint paths that start ``horizon`` behind the vehicle.  From the dense rows: the first ``rw_lanes`` lanes
    and the first ``rw_paths`` paths of each (include/smx.h)."""

    lanes: Dict[str, List[List[Waypoint]]]


@dataclass(frozen=True)
class EndlessMission:
    """What ``hiway-v0`` assigns an agent without ``missions.pkl`` (scenario.py:289-290,
    plan.py:76-84, 209): endless goal, empty route, start time 0.1 s."""

    start_time: float = 0.1
    goal: str = "EndlessGoal"
    route_roads: Tuple[str, ...] = ()


@dataclass(frozen=True)
class PositionalGoal:
    """plan.py:86-120."""

    position: Tuple[float, float]
    radius: float


@dataclass(frozen=True)
class FixedRouteMission:
    """plan.py:196-222 for a mission with a route: what ``EgoVehicleObservation.mission`` shows of it."""

    start_position: Tuple[float, float]
    start_heading: float
    goal: PositionalGoal
    route_roads: Tuple[str, ...]
    start_time: float = 0.1


class ObservationBuilder:
    """Dense rows (host numpy, one env) -> reference ``Observation`` objects."""

    def __init__(self, lane_ids: Sequence[str], lane_road_ids: Sequence[str], agent_ids: Sequence[str], *,
                 waypoints: bool, neighbors: bool, accelerometer: bool, ogm=None, lidar_rays: Optional[np.ndarray] = None,
                 dt: float = 0.1, vias=None, dagm=None, road_waypoints: bool = False, missions=None):
        self.lane_ids = list(lane_ids)
        self.lane_road_ids = list(lane_road_ids)
        self.agent_ids = list(agent_ids)
        self.waypoints, self.neighbors, self.accelerometer = waypoints, neighbors, accelerometer
        self.ogm, self.lidar_rays, self.dt = ogm, lidar_rays, dt
        self.dagm = dagm
        self.road_waypoints = road_waypoints
        # per vehicle slot: smarts_amd.missions.PlannedMission | None
        self.missions = [
            FixedRouteMission(tuple(m.start_position), float(m.start_heading), PositionalGoal(tuple(m.goal[:2]), float(m.goal[2])),
                              tuple(m.route_roads)) if m is not None else None for m in (missions or [])]
        self.vias = vias  # per vehicle slot: resolved mission vias (smarts_amd.vias.ResolvedVia)

    def vehicle_id(self, slot: int) -> str:
        # Vehicle.build_agent_vehicle (vehicle.py:371-372) names agent vehicles after their agent;
        # scripted social vehicles (slots after the agents) keep their own name
        name = self.agent_ids[slot]
        return name if name.startswith("social-") else f"{name}-vehicle"

    def _lane(self, lane: int, lane_index: int):
        if lane < 0:
            return None, None, None
        return self.lane_road_ids[lane], self.lane_ids[lane], int(lane_index)

    def build(self, rows: Dict[str, np.ndarray], slot: int, step_count: int, elapsed_sim_time: float,
              low_dimensional: bool = False) -> Observation:
        """``rows[k]`` is the [N, ...] slice of one env.  ``low_dimensional``: the ego block and the events only (the
        finishing tick of an env that restarted inside the launch: its sensor rows already hold the next episode)."""
        if low_dimensional:
            rows = dict(rows)
            for k in ("wp_count", "nb_count", "rw_lane", "via_near_count", "collidees"):
                if k in rows:
                    rows[k] = np.zeros_like(rows[k]) if k != "rw_lane" else np.full_like(rows[k], -1)
            saved = (self.ogm, self.dagm, self.lidar_rays)
            self.ogm = self.dagm = self.lidar_rays = None
            try:
                return self.build(rows, slot, step_count, elapsed_sim_time)
            finally:
                self.ogm, self.dagm, self.lidar_rays = saved
        E = nat.EGO
        f = rows["ego_f32"][slot]
        v3 = lambda k: np.array(f[E[k]:E[k] + 3], dtype=np.float64)  # noqa: E731
        road_id, lane_id, lane_index = self._lane(int(rows["ego_lane"][slot, 0]), int(rows["ego_lane"][slot, 1]))
        acc = self.accelerometer
        ego = EgoVehicleObservation(
            id=self.vehicle_id(slot),
            position=np.array(rows["ego_pos"][slot], dtype=np.float64),
            bounding_box=Dimensions(*[float(x) for x in f[E["BOX"]:E["BOX"] + 3]]),
            heading=Heading(float(f[E["HEADING"]])),
            speed=float(f[E["SPEED"]]),
            steering=float(f[E["STEERING"]]),
            yaw_rate=float(f[E["YAW_RATE"]]),
            road_id=road_id, lane_id=lane_id, lane_index=lane_index,
            mission=(self.missions[slot] if slot < len(self.missions) and self.missions[slot] is not None else EndlessMission()),
            linear_velocity=v3("LIN_VEL"), angular_velocity=v3("ANG_VEL"),
            linear_acceleration=v3("LIN_ACC") if acc else None,
            angular_acceleration=v3("ANG_ACC") if acc else None,
            linear_jerk=v3("LIN_JERK") if acc else None,
            angular_jerk=v3("ANG_JERK") if acc else None,
        )
        ev = rows["events"][slot]
        # _process_collisions walks the set of collidee body ids; bodies are created in slot order, so the
        # collisions come in slot order, each naming the owner of the vehicle hit (an agent id, or the
        # scripted social vehicle's own name)
        mask = int(rows["collidees"][slot]) & 0xFFFFFFFFFFFFFFFF if "collidees" in rows else 0
        collisions = [Collision(collidee_id=self.agent_ids[j]) for j in range(len(self.agent_ids)) if (mask >> j) & 1]
        if ev[nat.EV["COLLISIONS"]] and not collisions:  # rows from a caller that passed no collidee buffer
            collisions = [Collision(collidee_id=None)]
        events = Events(
            collisions=collisions,
            off_road=bool(ev[nat.EV["OFF_ROAD"]]), off_route=bool(ev[nat.EV["OFF_ROUTE"]]),
            on_shoulder=bool(ev[nat.EV["ON_SHOULDER"]]), wrong_way=bool(ev[nat.EV["WRONG_WAY"]]),
            not_moving=bool(ev[nat.EV["NOT_MOVING"]]), reached_goal=bool(ev[nat.EV["REACHED_GOAL"]]),
            reached_max_episode_steps=bool(ev[nat.EV["REACHED_MAX_EPISODE_STEPS"]]),
            agents_alive_done=bool(ev[nat.EV["AGENTS_ALIVE_DONE"]]),
        )
        neighbors = None
        if self.neighbors:
            neighbors = []
            for k in range(min(int(rows["nb_count"][slot]), rows["nb_slot"].shape[1])):
                r, l, li = self._lane(int(rows["nb_lane_id"][slot, k]), int(rows["nb_lane_index"][slot, k]))
                neighbors.append(VehicleObservation(
                    id=self.vehicle_id(int(rows["nb_slot"][slot, k])),
                    position=tuple(float(x) for x in rows["nb_pos"][slot, k]),
                    bounding_box=Dimensions(*[float(x) for x in rows["nb_box"][slot, k]]),
                    heading=Heading(float(rows["nb_heading"][slot, k])), speed=float(rows["nb_speed"][slot, k]),
                    road_id=r, lane_id=l, lane_index=li))
        paths = None
        if self.waypoints:
            paths = []
            counts = rows["wp_count"][slot]
            for p in range(min(int(counts[0]), len(counts) - 1)):
                path = []
                for w in range(int(counts[1 + p])):
                    lane = int(rows["wp_lane_id"][slot, p, w])
                    path.append(Waypoint(
                        pos=np.array(rows["wp_pos"][slot, p, w, :2], dtype=np.float64),
                        heading=Heading(float(rows["wp_heading"][slot, p, w])), lane_id=self.lane_ids[lane],
                        lane_width=float(rows["wp_lane_width"][slot, p, w]),
                        speed_limit=float(rows["wp_speed_limit"][slot, p, w]),
                        lane_index=int(rows["wp_lane_index"][slot, p, w])))
                paths.append(path)
        road_wps = None
        if self.road_waypoints:
            lanes = {}
            for l, lane in enumerate(rows["rw_lane"][slot]):
                if lane < 0:
                    continue
                lane_paths = []
                for p, n in enumerate(rows["rw_count"][slot, l]):
                    if p >= int(rows["rw_path_count"][slot, l]):
                        break
                    lane_paths.append([Waypoint(
                        pos=np.array(rows["rw_pos"][slot, l, p, w, :2], dtype=np.float64),
                        heading=Heading(float(rows["rw_heading"][slot, l, p, w])),
                        lane_id=self.lane_ids[int(rows["rw_lane_id"][slot, l, p, w])],
                        lane_width=float(rows["rw_lane_width"][slot, l, p, w]),
                        speed_limit=float(rows["rw_speed_limit"][slot, l, p, w]),
                        lane_index=int(rows["rw_lane_index"][slot, l, p, w])) for w in range(int(n))])
                lanes[self.lane_ids[int(lane)]] = lane_paths
            road_wps = RoadWaypoints(lanes=lanes)
        ogm = None
        if self.ogm is not None:
            meta = GridMapMetadata(
                created_at=int(elapsed_sim_time), resolution=self.ogm.resolution, width=self.ogm.width,
                height=self.ogm.height, camera_pos=tuple(float(x) for x in rows["ego_pos"][slot]),
                camera_heading_in_degrees=float(np.degrees(float(f[E["HEADING"]]))))
            ogm = OccupancyGridMap(metadata=meta, data=np.array(rows["ogm"][slot], dtype=np.uint8)[..., None])
        dagm = None
        if self.dagm is not None:
            meta = GridMapMetadata(
                created_at=int(elapsed_sim_time), resolution=self.dagm.resolution, width=self.dagm.width,
                height=self.dagm.height, camera_pos=tuple(float(x) for x in rows["ego_pos"][slot]),
                camera_heading_in_degrees=float(np.degrees(float(f[E["HEADING"]]))))
            dagm = DrivableAreaGridMap(metadata=meta, data=np.array(rows["dagm"][slot], dtype=np.uint8)[..., None])
        lidar = None
        if self.lidar_rays is not None:
            origin = np.array(rows["ego_pos"][slot], dtype=np.float64) + np.array([0.0, 0.0, 1.0])
            hits = [bool(h) for h in rows["lidar_hit"][slot]]
            pts = [np.array(p, dtype=np.float64) for p in rows["lidar_point"][slot]]
            rays = [(origin.copy(), origin + d) for d in self.lidar_rays]  # lidar.py:109-113
            lidar = (pts, hits, rays)
        via_data = Vias(near_via_points=[], hit_via_points=[])
        if self.vias is not None and self.vias[slot] and "via_near" in rows:
            mine = self.vias[slot]

            def point(k):
                v = mine[k]
                return ViaPoint(position=tuple(v.position), lane_index=v.lane_index, road_id=v.road_id,
                                required_speed=v.required_speed)

            near = [point(int(k)) for k in rows["via_near"][slot] if k >= 0]
            hit = [point(k) for k in range(len(mine)) if int(rows["via_hit"][slot]) >> k & 1]
            via_data = Vias(near_via_points=near, hit_via_points=hit)
        return Observation(
            dt=self.dt, step_count=step_count, elapsed_sim_time=elapsed_sim_time, events=events, ego_vehicle_state=ego,
            neighborhood_vehicle_states=neighbors, waypoint_paths=paths, distance_travelled=float(rows["dist"][slot]),
            lidar_point_cloud=lidar, drivable_area_grid_map=dagm, occupancy_grid_map=ogm, top_down_rgb=None,
            road_waypoints=road_wps, via_data=via_data)
