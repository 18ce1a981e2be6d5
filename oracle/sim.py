"""Oracle: one environment instance, ticked the way ``SMARTS._step`` does (test infrastructure).

Restates, per agent and sequentially:

* the tick order of ``SMARTS._step`` (``smarts/core/smarts.py:236-327``): clock →
  agent actions/controllers → physics → collisions → sensors → observe → teardown;
* ``Sensors.observe`` and the event logic (``smarts/core/sensors.py:238-594``);
* ``TripMeterSensor`` / ``DrivenPathSensor`` / ``AccelerometerSensor``
  (``sensors.py:830-947, 1046-1087``), reward/score (``agent_manager.py:233-234``);
* ``SMARTS.neighborhood_vehicles_around_vehicle`` (``smarts.py:1191-1208``);
* ``_process_collisions`` (``smarts.py:1270-1291``) over
  ``_query_bullet_contact_points`` (``chassis.py:60-80``).  The pybullet AABB /
  closest-point queries are substituted by 2-D oriented-box distance <= 0.05 m of
  the chassis footprints (wheels lie inside the footprint); DESIGN.md "Substitutions".

Agents are Ackermann vehicles with the Lane action space, EndlessGoal missions and
an empty route (what ``hiway-v0`` gives agents of a scenario without
``missions.pkl``: ``scenario.py:289-290``, ``plan.py:225-249,321-323``).
"""
import math
from collections import deque
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import controller as ctl
from . import ref_math as rm
from . import sensors_extra as sx
from .dynamics import VehicleBody

COLLISION_LEEWAY = 0.05  # chassis.py:75-78
CHASSIS_LENGTH = 3.68  # vehicle.py:101 (passenger)


@dataclass
class AgentConfig:
    """The slice of ``AgentInterface`` this path reads (agent_interface.py:175-297)."""

    waypoints_lookahead: Optional[int] = 32  # None = sensor disabled
    neighborhood_radius: Optional[float] = None
    neighborhood_enabled: bool = False
    accelerometer: bool = True
    max_episode_steps: Optional[int] = None
    done_collision: bool = True
    done_off_road: bool = True
    done_off_route: bool = True
    done_on_shoulder: bool = False
    done_wrong_way: bool = False
    done_not_moving: bool = False
    not_moving_time: float = 60
    not_moving_distance: float = 1
    alive_min_ego: Optional[int] = None  # DoneCriteria.agents_alive (agent_interface.py:155-176)
    alive_min_total: Optional[int] = None
    alive_lists: tuple = ()  # ((agent slots), minimum alive) pairs
    action_space: str = "Lane"  # Lane | Continuous | ActuatorDynamic | LaneWithContinuousSpeed
    ogm: Optional[tuple] = None  # (width, height, resolution) — OGM (agent_interface.py:42-51)
    dagm: Optional[tuple] = None  # (width, height, resolution) — DrivableAreaGridMap (agent_interface.py:29-38)
    lidar_rays: Optional[np.ndarray] = None  # base rays [R, 3] (sensors_extra.base_rays)
    road_waypoints_horizon: Optional[int] = None  # RoadWaypoints.horizon (agent_interface.py); None = sensor off


class _Agent:
    def __init__(self, body, cfg, road_map, sim):
        self.body = body
        self.cfg = cfg
        self.alive = True
        self.steps = 0
        nl = road_map.nearest_lane(body.position)  # ControllerState.from_action_space (controllers/__init__.py:166-180)
        self.ctrl = ctl.LaneFollowingControllerState(nl.lane_id if nl else None)
        # TripMeterSensor.__init__ (sensors.py:885-898)
        self.wps_for_distance = []
        self.dist_travelled = 0.0
        self.last_dist_travelled = 0.0
        wps = road_map.waypoint_paths(body.position, body.heading, lookahead=1, within_radius=body.length)
        if wps:
            self.wps_for_distance.append(wps[0][0])
        self.driven_path = deque(maxlen=500)  # sensors.py:838
        self.linear_velocities = deque(maxlen=3)
        self.angular_velocities = deque(maxlen=3)
        self.collisions = []


def _box_corners(x, y, heading, length, width):
    f = (-math.sin(heading), math.cos(heading))
    r = (math.cos(heading), math.sin(heading))
    hl, hw = 0.5 * length, 0.5 * width
    return [
        (x + f[0] * hl + r[0] * hw, y + f[1] * hl + r[1] * hw),
        (x + f[0] * hl - r[0] * hw, y + f[1] * hl - r[1] * hw),
        (x - f[0] * hl - r[0] * hw, y - f[1] * hl - r[1] * hw),
        (x - f[0] * hl + r[0] * hw, y - f[1] * hl + r[1] * hw),
    ]


def _seg_point_dist2(px, py, ax, ay, bx, by):
    dx, dy = bx - ax, by - ay
    ll = dx * dx + dy * dy
    t = 0.0 if ll == 0.0 else ((px - ax) * dx + (py - ay) * dy) / ll
    t = min(1.0, max(0.0, t))
    ex, ey = ax + t * dx - px, ay + t * dy - py
    return ex * ex + ey * ey


def _point_in_box(px, py, x, y, heading, length, width):
    f = (-math.sin(heading), math.cos(heading))
    r = (math.cos(heading), math.sin(heading))
    dx, dy = px - x, py - y
    return abs(dx * f[0] + dy * f[1]) <= 0.5 * length and abs(dx * r[0] + dy * r[1]) <= 0.5 * width


def boxes_within(a, b, leeway):
    """True iff the 2-D distance between two oriented footprints is <= leeway.

    Distance between convex polygons: 0 if they intersect, else attained between a
    vertex of one and an edge of the other.
    """
    ca = _box_corners(a.x, a.y, a.heading, a.length, a.width)
    cb = _box_corners(b.x, b.y, b.heading, b.length, b.width)
    best = float("inf")
    for P, Q, other in ((ca, cb, b), (cb, ca, a)):
        for (px, py) in P:
            if _point_in_box(px, py, other.x, other.y, other.heading, other.length, other.width):
                return True
            for k in range(4):
                ax, ay = Q[k]
                bx, by = Q[(k + 1) % 4]
                best = min(best, _seg_point_dist2(px, py, ax, ay, bx, by))
    if best <= leeway * leeway:
        return True
    # edge-edge crossings without any contained vertex (a "plus" configuration)
    for i in range(4):
        a0, a1 = ca[i], ca[(i + 1) % 4]
        for k in range(4):
            b0, b1 = cb[k], cb[(k + 1) % 4]
            d1 = (a1[0] - a0[0]) * (b0[1] - a0[1]) - (a1[1] - a0[1]) * (b0[0] - a0[0])
            d2 = (a1[0] - a0[0]) * (b1[1] - a0[1]) - (a1[1] - a0[1]) * (b1[0] - a0[0])
            d3 = (b1[0] - b0[0]) * (a0[1] - b0[1]) - (b1[1] - b0[1]) * (a0[0] - b0[0])
            d4 = (b1[0] - b0[0]) * (a1[1] - b0[1]) - (b1[1] - b0[1]) * (a1[0] - b0[0])
            if (d1 > 0) != (d2 > 0) and (d3 > 0) != (d4 > 0):
                return True
    return False


class SocialBody(VehicleBody):
    """Scripted social vehicle (the model of include/smx.h ``smx_config.num_social``; stands in for
    the SUMO provider's BoxChassis vehicles, smarts.py:868-921 — parity with SUMO itself is unpinned):
    follows its lane's centre line at ``factor`` x the speed limit, continues on outgoing lane
    ``(slot + lanes crossed) mod #outgoing``, stops at the end of a lane without successors."""

    def __init__(self, x, y, heading, speed, lane, offset, slot, factor):
        super().__init__(x, y, heading, speed)
        self.lane, self.offset, self.slot, self.factor, self.crossed = lane, float(offset), slot, factor, 0
        self.speed_cmd = None  # set every tick by the IDM model; None = constant fraction of the limit

    @staticmethod
    def _cum(lane):
        acc, cum = 0.0, [0.0]
        for a, b in zip(lane.shape[:-1], lane.shape[1:]):
            ex, ey = float(a[0] - b[0]), float(a[1] - b[1])
            acc = acc + math.sqrt(ex * ex + ey * ey)
            cum.append(acc)
        return cum

    def control(self, *a, **k):
        pass

    # IDM car following (include/smx.h SMX_SOCIAL_IDM; SUMO's passenger defaults accel 2.6, decel 4.5,
    # tau 1.0, minGap 2.5 — the arithmetic below is this project's own statement of the model)
    IDM_ACCEL, IDM_DECEL, IDM_TAU, IDM_MIN_GAP = 2.6, 4.5, 1.0, 2.5
    IDM_HORIZON, IDM_CORRIDOR = 60.0, 1.6

    def idm_speed(self, others, dt):
        """Speed for the coming tick from the poses / speeds at the start of the tick.
        ``others``: (slot, body) of every other alive vehicle, in slot order."""
        v = self.u
        v0 = self.lane.speed_limit * self.factor
        fx, fy = -math.sin(self.heading), math.cos(self.heading)
        rx, ry = math.cos(self.heading), math.sin(self.heading)
        best, lead_u, found = self.IDM_HORIZON, 0.0, False
        for _, b in others:
            dx, dy = b.x - self.x, b.y - self.y
            lon = dx * fx + dy * fy
            lat = dx * rx + dy * ry
            if lon > 0.0 and lon < best and abs(lat) < self.IDM_CORRIDOR:
                best, lead_u, found = lon, b.u, True
        if v0 <= 0.0:
            return max(0.0, v - self.IDM_DECEL * dt)
        ratio = v / v0
        free = 1.0 - (ratio * ratio) * (ratio * ratio)
        inter = 0.0
        if found:
            gap = max(best - CHASSIS_LENGTH, 0.1)
            dv = v - lead_u
            sstar = self.IDM_MIN_GAP + max(0.0, v * self.IDM_TAU + v * dv / (2.0 * math.sqrt(self.IDM_ACCEL * self.IDM_DECEL)))
            q = sstar / gap
            inter = q * q
        acc = self.IDM_ACCEL * (free - inter)
        return min(max(v + acc * dt, 0.0), v0)

    def step(self, dt):
        speed = self.lane.speed_limit * self.factor if self.speed_cmd is None else self.speed_cmd
        self.offset += speed * dt
        for _ in range(64):
            L = self._cum(self.lane)[-1]
            if self.offset < L:
                break
            outs = self.lane.outgoing_lanes
            if not outs:
                self.offset, speed = L, 0.0
                break
            self.offset -= L
            self.lane = outs[(self.slot + self.crossed) % len(outs)]
            self.crossed += 1
        shape, cum = self.lane.shape, self._cum(self.lane)
        seg = len(shape) - 2
        for v in range(len(shape) - 1):
            if cum[v] + self._seg_len(shape, v) > self.offset:
                seg = v
                break
        a, b = shape[seg], shape[seg + 1]
        ln = self._seg_len(shape, seg)
        along = min(max(self.offset - cum[seg], 0.0), ln)
        f = along / ln if ln > 0.0 else 0.0
        self.x = a[0] + (b[0] - a[0]) * f
        self.y = a[1] + (b[1] - a[1]) * f
        self.heading = rm.wrap_heading(math.atan2(b[1] - a[1], b[0] - a[0]) - 0.5 * math.pi)
        self.u, self.v, self.yaw_rate_z = speed, 0.0, 0.0

    @staticmethod
    def _seg_len(shape, v):
        ex, ey = float(shape[v][0] - shape[v + 1][0]), float(shape[v][1] - shape[v + 1][1])
        return math.sqrt(ex * ex + ey * ey)


class _Social:
    alive = True
    collisions = ()

    def __init__(self, body):
        self.body = body


class OracleEnv:
    """One SMARTS instance with N ego agents (and optional scripted social vehicles) on one map."""

    def __init__(self, road_map, spawns, configs, dt=0.1, social=(), social_speed_factor=0.8, vias=None,
                 social_model="constant", missions=None):
        """``spawns``: (N, 4) array of x, y, heading, speed (vehicle centre) for the agents followed
        by the social vehicles; ``social``: (lane id, arclength offset) per social vehicle;
        ``missions``: per agent ``None`` (endless mission, empty route: plan.py:321-323) or
        ``dict(route=[road ids], goal=(x, y, radius))`` — a fixed-route mission with a
        ``PositionalGoal`` (plan.py:86-120, 316-349; the route as ``ORoadNetwork.create_route`` plans it)."""
        self.road_map = road_map
        self.dt = dt
        self._round = rm.round_param_for_dt(dt)
        # SMARTS.reset spins step({}) until the traps have fired (smarts.py:426-434): the ego
        # vehicles appear once mission.start_time = 0.1 s has *passed* (trap_manager.py:53-65)
        self.reset_steps = int(math.floor(0.1 / dt + 1e-9)) + 1
        self.elapsed_sim_time = 0.0
        for _ in range(self.reset_steps):
            self.elapsed_sim_time = round(self.elapsed_sim_time + dt, self._round)
        self.step_count = self.reset_steps - 1
        spawns = np.asarray(spawns, dtype=np.float64)
        n_agents = len(spawns) - len(social)
        self.agents = [_Agent(VehicleBody(*s), c, road_map, self) for s, c in zip(spawns[:n_agents], configs)]
        # per agent: the mission's vias as dicts(lane_id, position, hit_distance, required_speed, ...)
        self.vias = vias if vias is not None else [[] for _ in self.agents]
        missions = missions if missions is not None else [None] * len(self.agents)
        for ag, mission in zip(self.agents, missions):
            ag.consumed_vias = set()
            ag.route = tuple(mission["route"]) if mission else ()
            ag.goal = tuple(mission["goal"]) if mission else None
        self.social = [
            _Social(SocialBody(*spawns[n_agents + k], road_map.lane_by_id(lane_id), off, n_agents + k, social_speed_factor))
            for k, (lane_id, off) in enumerate(social)
        ]
        self.social_model = social_model

    def _vehicles(self):
        """(slot, body) of every vehicle in the world, agents first (the vehicle-index order)."""
        return [(j, v.body) for j, v in enumerate(list(self.agents) + list(self.social)) if v.alive]

    def reset_observe(self):
        """The observations ``SMARTS.reset`` returns: sensors run on the just-created vehicles
        (no controller, no physics yet)."""
        alive_states = self._vehicles()
        obs = {}
        for i, ag in enumerate(self.agents):
            ag.steps += 1
            obs[i], _ = self._observe(i, ag, alive_states)
        self.step_count += 1
        return obs

    # ------------------------------------------------------------------ tick
    def step(self, actions):
        """``actions[i]`` is a Lane action name (or index into LANE_ACTION_NAMES) or None."""
        rmap = self.road_map
        self.elapsed_sim_time = round(self.elapsed_sim_time + self.dt, self._round)  # smarts.py:261-262
        if self.social_model == "idm":
            # every social vehicle decides from the state at the start of the tick
            everyone = self._vehicles()
            n_agents = len(self.agents)
            cmds = [sv.body.idm_speed([(j, b) for j, b in everyone if j != n_agents + k], self.dt)
                    for k, sv in enumerate(self.social)]
            for sv, c in zip(self.social, cmds):
                sv.body.speed_cmd = c
        # 2. controllers (smarts.py:1233-1263)
        for ag, action in zip(self.agents, actions):
            if not ag.alive:
                continue
            space = ag.cfg.action_space
            none = action is None
            if not none and space == "Lane" and not isinstance(action, str) and int(action) < 0:
                none = True
            if not none and space == "Trajectory" and len(action[0]) == 0:
                none = True
            if not none and space not in ("Lane", "Trajectory") and np.isnan(float(action[0])):
                none = True
            if none:
                # no action this tick (controllers/__init__.py:87-88): wheel torques last one
                # physics step only, the steer motor keeps its target
                ag.body.control(throttle=0.0, brake=0.0, steering=ag.ctrl.steering_state)
                continue
            if space == "Trajectory":  # controllers/__init__.py:104-110
                if not isinstance(ag.ctrl, ctl.TrajectoryTrackingControllerState):
                    ag.ctrl = ctl.TrajectoryTrackingControllerState()  # ControllerState.from_action_space
                thr, brk, steer = ctl.perform_trajectory_tracking_pd(action, ag.body, ag.ctrl, self.dt)
            elif space == "Continuous":  # controllers/__init__.py:94-99
                thr, brk, steer = (float(np.clip(action[0], 0.0, 1.0)), float(np.clip(action[1], 0.0, 1.0)),
                                   float(np.clip(action[2], -1, 1)))
                ag.ctrl.steering_state = steer
            elif space == "ActuatorDynamic":  # actuator_dynamic_controller.py:47-80
                change = float(np.clip(action[2], -1, 1))
                p = 0.001
                steer = float(np.clip((1 - p) * ag.ctrl.steering_state + change * self.dt, -1, 1))
                thr, brk = float(np.clip(action[0], 0.0, 1.0)), float(np.clip(action[1], 0.0, 1.0))
                ag.ctrl.steering_state = steer  # last_steering_angle
            else:
                if space == "LaneWithContinuousSpeed":  # controllers/__init__.py:113-124
                    target_speed, lane_change = float(action[0]), int(action[1])
                else:
                    if not isinstance(action, str):
                        action = ctl.LANE_ACTION_NAMES[int(action)]
                    target_speed, lane_change = ctl.LANE_ACTIONS[action]
                thr, brk, steer = ctl.perform_lane_following(
                    rmap, ag.body, ag.ctrl, self.dt, target_speed=target_speed, lane_change=lane_change, route=ag.route
                )
            ag.body.control(throttle=thr, brake=brk, steering=steer)
        # physics (smarts.py:923-931)
        for ag in self.agents:
            if ag.alive:
                ag.body.step(self.dt)
        for sv in self.social:
            sv.body.step(self.dt)
        # collisions (smarts.py:1270-1291)
        everyone = self._vehicles()
        for i, ag in enumerate(self.agents):
            ag.collisions = []
            if not ag.alive:
                continue
            for j, other in everyone:
                if j == i:
                    continue
                if boxes_within(ag.body, other, COLLISION_LEEWAY):
                    ag.collisions.append(j)
        # sensors + observe (smarts.py:287-301)
        alive_states = everyone
        obs, rewards, dones = {}, {}, {}
        for i, ag in enumerate(self.agents):
            if not ag.alive:
                continue
            ag.steps += 1  # SensorState.step (agent_manager.py:250-258)
            o, done = self._observe(i, ag, alive_states)
            obs[i] = o
            dones[i] = done
            rewards[i] = ag.dist_travelled - ag.last_dist_travelled  # agent_manager.py:233
        for i, d in dones.items():  # smarts.py:314
            if d:
                self.agents[i].alive = False
        self.step_count += 1
        return obs, rewards, dones

    # ------------------------------------------------------------------ observe
    def _observe(self, i, ag, alive_states):
        rmap, b, cfg = self.road_map, ag.body, ag.cfg
        o = {}
        # neighbourhood (sensors.py:241-266, smarts.py:1191-1208)
        if cfg.neighborhood_enabled:
            nvs = []
            for j, ob in alive_states:
                if j == i:
                    continue
                if cfg.neighborhood_radius is not None:
                    d = np.linalg.norm(ob.position - b.position)
                    if not d <= cfg.neighborhood_radius:
                        continue
                nv_lane = rmap.nearest_lane(ob.position, radius=b.length)
                nvs.append(
                    dict(
                        slot=j,
                        position=ob.position,
                        box=(ob.length, ob.width, ob.height),
                        heading=ob.heading,
                        speed=ob.speed,
                        lane_id=nv_lane.lane_id if nv_lane else None,
                        road_id=nv_lane.road.road_id if nv_lane else None,
                        lane_index=nv_lane.index if nv_lane else None,
                    )
                )
            o["neighbors"] = nvs
        # waypoints (sensors.py:268-275, 972-985)
        if cfg.waypoints_lookahead is not None:
            waypoint_paths = rmap.waypoint_paths(b.position, b.heading, lookahead=cfg.waypoints_lookahead, route=ag.route)
        else:
            waypoint_paths = rmap.waypoint_paths(b.position, b.heading, lookahead=1, within_radius=b.length)
        closest_lane = rmap.nearest_lane(b.position)
        lin_v, ang_v = b.linear_velocity, b.angular_velocity
        o["ego"] = dict(
            position=np.array(b.position),
            box=(b.length, b.width, b.height),
            heading=rm.wrap_heading(b.heading),
            speed=b.speed,
            steering=b.steering,
            yaw_rate=b.yaw_rate,
            lane_id=closest_lane.lane_id if closest_lane else None,
            road_id=closest_lane.road.road_id if closest_lane else None,
            lane_index=closest_lane.index if closest_lane else None,
            linear_velocity=lin_v,
            angular_velocity=ang_v,
        )
        if cfg.accelerometer:
            o["ego"].update(self._accelerometer(ag, lin_v, ang_v))
        # trip meter (sensors.py:349-351, 900-944)
        if waypoint_paths:
            self._append_waypoint_if_new(ag, waypoint_paths[0][0])
        o["distance_travelled"] = ag.dist_travelled
        ag.driven_path.append((self.elapsed_sim_time, b.position[:2]))  # sensors.py:842-847
        o["waypoint_paths"] = waypoint_paths if cfg.waypoints_lookahead is not None else None
        if cfg.ogm is not None:  # sensors.py:303-305
            o["ogm"] = sx.ogm(b, [ob for _, ob in alive_states], *cfg.ogm)
        if cfg.dagm is not None:  # sensors.py:307-312
            o["dagm"] = sx.dagm(b, rmap.lane_bands(), *cfg.dagm)
        if cfg.road_waypoints_horizon is not None:  # sensors.py:277-281, 991-1040
            o["road_waypoints"] = sx.road_waypoints(rmap, b.position, b.heading, cfg.road_waypoints_horizon, ag.route)
        if cfg.lidar_rays is not None:  # sensors.py:297-301
            o["lidar"] = sx.lidar(b, [ob for j, ob in alive_states if j != i], cfg.lidar_rays)
        if self.vias[i]:
            o["vias"] = self._via_sensor(ag, self.vias[i])
        done, events = self._is_done_with_events(ag)
        o["events"] = events
        o["dt"] = self.dt
        o["step_count"] = self.step_count
        o["elapsed_sim_time"] = self.elapsed_sim_time
        return o, done

    def _accelerometer(self, ag, linear_velocity, angular_velocity):
        """sensors.py:1053-1084."""
        dt = self.dt
        ag.linear_velocities.append(linear_velocity)
        ag.angular_velocities.append(angular_velocity)
        la = np.array((0.0, 0.0, 0.0))
        aa = np.array((0.0, 0.0, 0.0))
        lj = np.array((0.0, 0.0, 0.0))
        aj = np.array((0.0, 0.0, 0.0))
        if len(ag.linear_velocities) >= 2:
            la = (ag.linear_velocities[-1] - ag.linear_velocities[-2]) / dt
            if len(ag.linear_velocities) >= 3:
                lj = la - (ag.linear_velocities[-2] - ag.linear_velocities[-3]) / dt
        if len(ag.angular_velocities) >= 2:
            aa = (ag.angular_velocities[-1] - ag.angular_velocities[-2]) / dt
            if len(ag.angular_velocities) >= 3:
                aj = aa - (ag.angular_velocities[-2] - ag.angular_velocities[-3]) / dt
        return dict(linear_acceleration=la, angular_acceleration=aa, linear_jerk=lj, angular_jerk=aj)

    def _append_waypoint_if_new(self, ag, new_wp):
        """sensors.py:900-938."""
        ag.last_dist_travelled = ag.dist_travelled
        wp_road = self.road_map.lane_by_id(new_wp.lane_id).road.road_id
        # an endless mission counts every waypoint, a fixed route only those on its roads (:908-913)
        should_count_wp = ag.goal is None or wp_road in ag.route
        if not ag.wps_for_distance:
            if should_count_wp:
                ag.wps_for_distance.append(new_wp)
            return
        recent = ag.wps_for_distance[-1]
        if np.linalg.norm(new_wp.pos - recent.pos) > 0.5 and should_count_wp:
            heading_vec = rm.radians_to_vec(recent.heading)
            disp_vec = new_wp.pos - recent.pos
            direction = np.sign(np.dot(heading_vec, disp_vec))
            ag.dist_travelled += direction * np.linalg.norm(disp_vec)
            ag.wps_for_distance.append(new_wp)

    # ------------------------------------------------------------------ events
    def _is_done_with_events(self, ag):
        """sensors.py:443-489."""
        rmap, b, cfg = self.road_map, ag.body, ag.cfg
        reached_goal = False  # EndlessGoal (plan.py:76-84)
        if ag.goal is not None:  # PositionalGoal.is_reached (plan.py:116-120) via Mission.is_complete (:220-222)
            sqr_dist = (b.position[0] - ag.goal[0]) ** 2 + (b.position[1] - ag.goal[1]) ** 2
            reached_goal = bool(sqr_dist <= ag.goal[2] ** 2)
        collided = len(ag.collisions) > 0
        is_off_road = not rmap.road_with_point(b.position)  # sensors.py:498-500
        is_on_shoulder = False  # sensors.py:502-509
        for corner in b.bounding_box:
            if not rmap.road_with_point((corner[0], corner[1], 0)):
                is_on_shoulder = True
                break
        is_not_moving = self._not_moving(ag)
        reached_max = cfg.max_episode_steps is not None and ag.steps >= cfg.max_episode_steps
        is_off_route, is_wrong_way = self._off_route_and_wrong_way(b, ag.route)
        agents_alive_done = self._agents_alive_done(cfg)
        done = (
            agents_alive_done
            or (is_off_road and cfg.done_off_road)
            or reached_goal
            or reached_max
            or (is_on_shoulder and cfg.done_on_shoulder)
            or (collided and cfg.done_collision)
            or (is_not_moving and cfg.done_not_moving)
            or (is_off_route and cfg.done_off_route)
            or (is_wrong_way and cfg.done_wrong_way)
        )
        events = dict(
            collisions=list(ag.collisions),
            off_road=is_off_road,
            reached_goal=reached_goal,
            reached_max_episode_steps=reached_max,
            off_route=is_off_route,
            on_shoulder=is_on_shoulder,
            wrong_way=is_wrong_way,
            not_moving=is_not_moving,
            agents_alive_done=agents_alive_done,
        )
        return bool(done), events

    def _via_sensor(self, ag, vias, acquisition_range=40, speed_accuracy=1.5):
        """ViaSensor.__call__ (sensors.py:1103-1146; range / tolerance from vehicle.py:553-557):
        (indices of the near vias, nearest first; indices hit this tick)."""
        b = ag.body
        pos = np.array(b.position[:2])
        near, hit = [], []
        for k, via in enumerate(vias):
            lane = self.road_map.lane_by_id(via["lane_id"])
            centre = np.array(lane.from_lane_coord(lane.offset_along_lane(tuple(pos)))[:2])  # center_at_point
            delta = centre - pos
            if np.dot(delta, delta) > acquisition_range ** 2:
                continue
            near.append(k)
            dv = np.array(via["position"]) - pos
            if (np.dot(dv, dv) <= via["hit_distance"] ** 2 and k not in ag.consumed_vias
                    and np.isclose(b.speed, via["required_speed"], atol=speed_accuracy)):
                ag.consumed_vias.add(k)
                hit.append(k)

        def sq(k):
            d = pos - np.array(vias[k]["position"])  # squared_dist(point.position, vehicle_position)
            return np.dot(d, d)

        return sorted(near, key=sq), hit

    def _agents_alive_done(self, cfg):
        """sensors.py:404-441: ``agent_manager.agent_ids`` holds the agents not yet torn down, i.e.
        alive at the start of this tick (teardown follows the observations, smarts.py:314)."""
        ids = {i for i, a in enumerate(self.agents) if a.alive}
        if cfg.alive_min_ego and len(ids) < cfg.alive_min_ego:
            return True
        if cfg.alive_min_total and len(ids) < cfg.alive_min_total:
            return True
        for slots, minimum in cfg.alive_lists:
            if [1 if i in ids else 0 for i in slots].count(1) < minimum:
                return True
        return False

    def _not_moving(self, ag):
        """sensors.py:511-525, 855-877."""
        cfg = ag.cfg
        if self.elapsed_sim_time < cfg.not_moving_time:
            return False
        threshold = self.elapsed_sim_time - cfg.not_moving_time
        pts = [p for (t, p) in ag.driven_path if t >= threshold]
        xs = np.array([p[0] for p in pts])
        ys = np.array([p[1] for p in pts])
        dist_array = (xs[:-1] - xs[1:]) ** 2 + (ys[:-1] - ys[1:]) ** 2
        return bool(np.sum(np.sqrt(dist_array)) < cfg.not_moving_distance)

    def _off_route_and_wrong_way(self, b, route_roads=()):
        """sensors.py:527-594."""
        radius = np.linalg.norm((b.length, b.width)) * 0.5 + 5
        nearest_lane = self.road_map.nearest_lane(b.position, radius=radius)
        if not nearest_lane:
            return (True, False)
        if nearest_lane.in_junction:
            is_wrong_way = False
        else:
            target_heading = nearest_lane.center_pose_heading_at_point(tuple(b.position))
            is_wrong_way = bool(np.fabs(rm.heading_relative_to(b.heading, target_heading)) > 0.5 * np.pi)
        if not route_roads or nearest_lane.road.road_id in route_roads or nearest_lane.in_junction:
            return (False, is_wrong_way)
        # not on the route, but perhaps just over the centre line in an oncoming lane (:566-571)
        veh_offset = nearest_lane.offset_along_lane(tuple(b.position))
        for on_lane in nearest_lane.oncoming_lanes_at_offset(veh_offset):
            if on_lane.road.road_id in route_roads:
                return (False, is_wrong_way)
        return (True, is_wrong_way)
