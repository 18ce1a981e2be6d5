"""Shared helpers for parity tests: run the oracle on the same spawns/actions and pack its
ragged Python observations into the device's dense layout (include/smx.h)."""
import numpy as np

from oracle.road_network import ORoadNetwork
from oracle.sim import AgentConfig, OracleEnv
from smarts_amd import _native as nat


def oracle_config(cfg):
    return AgentConfig(
        waypoints_lookahead=cfg.wp_lookahead if cfg.waypoints else None,
        neighborhood_radius=cfg.nb_radius,
        neighborhood_enabled=cfg.neighbors,
        accelerometer=cfg.accelerometer,
        max_episode_steps=cfg.max_episode_steps,
        done_collision=cfg.done_collision,
        done_off_road=cfg.done_off_road,
        done_off_route=cfg.done_off_route,
        done_on_shoulder=cfg.done_on_shoulder,
        done_wrong_way=cfg.done_wrong_way,
        done_not_moving=cfg.done_not_moving,
        not_moving_time=cfg.not_moving_time,
        not_moving_distance=cfg.not_moving_distance,
        action_space=cfg.action_space,
        alive_min_ego=cfg.alive_min_ego, alive_min_total=cfg.alive_min_total,
        alive_lists=tuple((tuple(s_), m) for s_, m in cfg.alive_lists),
        ogm=(cfg.ogm_width, cfg.ogm_height, cfg.ogm_resolution) if cfg.ogm else None,
        dagm=(cfg.dagm_width, cfg.dagm_height, cfg.dagm_resolution) if cfg.dagm else None,
        lidar_rays=oracle_lidar_rays(cfg.lidar) if cfg.lidar is not None else None,
        road_waypoints_horizon=cfg.rw_horizon if cfg.road_waypoints else None,
    )


def oracle_lidar_rays(p):
    from oracle.sensors_extra import base_rays

    return base_rays(p.start_angle, p.end_angle, p.laser_angles, p.angle_resolution, p.max_distance)


def empty_dense(cfg, n):
    P, W, K = cfg.wp_paths, cfg.wp_len, cfg.nb_max
    d = dict(
        ego_pos=np.zeros((n, 3)), ego_f32=np.zeros((n, nat.EGO_F32_COUNT), np.float32),
        ego_lane=np.full((n, 2), -1, np.int16), events=np.zeros((n, nat.EV_COUNT), np.uint8),
        reward=np.zeros(n), dist=np.zeros(n), done=np.zeros(n, np.uint8), active=np.zeros(n, np.uint8),
        collidees=np.zeros(n, np.uint64),
    )
    if cfg.waypoints:
        d.update(
            wp_pos=np.zeros((n, P, W, 3)), wp_heading=np.zeros((n, P, W), np.float32),
            wp_lane_width=np.zeros((n, P, W), np.float32), wp_speed_limit=np.zeros((n, P, W), np.float32),
            wp_lane_index=np.zeros((n, P, W), np.int8), wp_lane_id=np.full((n, P, W), -1, np.int16),
            wp_count=np.zeros((n, P + 1), np.uint8),
        )
    if cfg.neighbors:
        d.update(
            nb_pos=np.zeros((n, K, 3)), nb_box=np.zeros((n, K, 3), np.float32), nb_heading=np.zeros((n, K), np.float32),
            nb_speed=np.zeros((n, K), np.float32), nb_lane_index=np.zeros((n, K), np.int8),
            nb_lane_id=np.full((n, K), -1, np.int16), nb_slot=np.full((n, K), -1, np.int8),
            nb_count=np.zeros(n, np.uint8),
        )
    if cfg.via_max > 0:
        d.update(via_near=np.full((n, cfg.via_max), -1, np.int8), via_near_count=np.zeros(n, np.uint8),
                 via_hit=np.zeros(n, np.int32))
    if cfg.ogm:
        d["ogm"] = np.zeros((n, cfg.ogm_height, cfg.ogm_width), np.uint8)
    if cfg.dagm:
        d["dagm"] = np.zeros((n, cfg.dagm_height, cfg.dagm_width), np.uint8)
    if cfg.road_waypoints:
        L, Q, R = cfg.rw_lanes, cfg.rw_paths, 2 * cfg.rw_horizon + 1
        d.update(
            rw_lane_count=np.zeros(n, np.uint8), rw_lane=np.full((n, L), -1, np.int16), rw_path_count=np.zeros((n, L), np.int16),
            rw_count=np.zeros((n, L, Q), np.uint8), rw_pos=np.zeros((n, L, Q, R, 3)), rw_heading=np.zeros((n, L, Q, R), np.float32),
            rw_lane_width=np.zeros((n, L, Q, R), np.float32), rw_speed_limit=np.zeros((n, L, Q, R), np.float32),
            rw_lane_index=np.zeros((n, L, Q, R), np.int8), rw_lane_id=np.zeros((n, L, Q, R), np.int16),
        )
    if cfg.lidar is not None:
        from smarts_amd.lidar import ray_count

        d["lidar_hit"] = np.zeros((n, ray_count(cfg.lidar)), np.uint8)
        d["lidar_point"] = np.zeros((n, ray_count(cfg.lidar), 3))
    return d


def pack(cfg, lane_no, n, obs, rewards=None, dones=None):
    """Oracle observation dicts of one env -> dense arrays (rows of absent agents stay zero)."""
    d = empty_dense(cfg, n)
    E = nat.EGO
    for i, o in obs.items():
        e = o["ego"]
        d["ego_pos"][i] = e["position"]
        f = d["ego_f32"][i]
        f[E["HEADING"]] = e["heading"]
        f[E["SPEED"]] = e["speed"]
        f[E["STEERING"]] = e["steering"]
        f[E["YAW_RATE"]] = e["yaw_rate"]
        f[E["LIN_VEL"]:E["LIN_VEL"] + 3] = e["linear_velocity"]
        f[E["ANG_VEL"]:E["ANG_VEL"] + 3] = e["angular_velocity"]
        if "linear_acceleration" in e:
            f[E["LIN_ACC"]:E["LIN_ACC"] + 3] = e["linear_acceleration"]
            f[E["ANG_ACC"]:E["ANG_ACC"] + 3] = e["angular_acceleration"]
            f[E["LIN_JERK"]:E["LIN_JERK"] + 3] = e["linear_jerk"]
            f[E["ANG_JERK"]:E["ANG_JERK"] + 3] = e["angular_jerk"]
        f[E["BOX"]:E["BOX"] + 3] = e["box"]
        d["ego_lane"][i, 0] = lane_no[e["lane_id"]] if e["lane_id"] is not None else -1
        d["ego_lane"][i, 1] = e["lane_index"] if e["lane_index"] is not None else -1
        ev = o["events"]
        d["events"][i] = [
            len(ev["collisions"]) > 0, ev["off_road"], ev["off_route"], ev["on_shoulder"], ev["wrong_way"],
            ev["not_moving"], ev["reached_goal"], ev["reached_max_episode_steps"], ev["agents_alive_done"],
        ]
        d["collidees"][i] = np.uint64(sum(1 << int(j) for j in ev["collisions"]))  # one bit per collidee slot
        d["dist"][i] = o["distance_travelled"]
        d["active"][i] = 1
        if rewards is not None:
            d["reward"][i] = rewards[i]
        if dones is not None:
            d["done"][i] = dones[i]
            d["active"][i] = 0 if dones[i] else 1
        if cfg.waypoints and o["waypoint_paths"] is not None:
            paths = o["waypoint_paths"]
            d["wp_count"][i, 0] = min(len(paths), 255)
            for p, path in enumerate(paths[: cfg.wp_paths]):
                d["wp_count"][i, 1 + p] = min(len(path), cfg.wp_len)
                for w, wp in enumerate(path[: cfg.wp_len]):
                    d["wp_pos"][i, p, w, :2] = wp.pos
                    d["wp_heading"][i, p, w] = wp.heading
                    d["wp_lane_width"][i, p, w] = wp.lane_width
                    d["wp_speed_limit"][i, p, w] = wp.speed_limit
                    d["wp_lane_index"][i, p, w] = wp.lane_index
                    d["wp_lane_id"][i, p, w] = lane_no[wp.lane_id]
        if cfg.neighbors:
            nvs = o["neighbors"]
            d["nb_count"][i] = min(len(nvs), 255)
            for k, nv in enumerate(nvs[: cfg.nb_max]):
                d["nb_pos"][i, k] = nv["position"]
                d["nb_box"][i, k] = nv["box"]
                d["nb_heading"][i, k] = nv["heading"]
                d["nb_speed"][i, k] = nv["speed"]
                d["nb_lane_index"][i, k] = nv["lane_index"] if nv["lane_index"] is not None else -1
                d["nb_lane_id"][i, k] = lane_no[nv["lane_id"]] if nv["lane_id"] is not None else -1
                d["nb_slot"][i, k] = nv["slot"]
        if cfg.via_max > 0 and "vias" in o:
            near, hit = o["vias"]
            d["via_near_count"][i] = min(len(near), 255)
            d["via_near"][i, :min(len(near), cfg.via_max)] = near[:cfg.via_max]
            d["via_hit"][i] = sum(1 << k for k in hit)
        if cfg.road_waypoints:
            lanes = o["road_waypoints"]  # {lane id: [paths]} in the reference's dict order
            d["rw_lane_count"][i] = min(len(lanes), 255)
            for l, (lane_id, paths) in enumerate(list(lanes.items())[: cfg.rw_lanes]):
                d["rw_lane"][i, l] = lane_no[lane_id]
                d["rw_path_count"][i, l] = min(len(paths), 32767)
                for p, path in enumerate(paths[: cfg.rw_paths]):
                    d["rw_count"][i, l, p] = len(path)
                    for w, wp in enumerate(path):
                        d["rw_pos"][i, l, p, w, :2] = wp.pos
                        d["rw_heading"][i, l, p, w] = wp.heading
                        d["rw_lane_width"][i, l, p, w] = wp.lane_width
                        d["rw_speed_limit"][i, l, p, w] = wp.speed_limit
                        d["rw_lane_index"][i, l, p, w] = wp.lane_index
                        d["rw_lane_id"][i, l, p, w] = lane_no[wp.lane_id]
        if cfg.ogm:
            d["ogm"][i] = o["ogm"]
        if cfg.dagm:
            d["dagm"][i] = o["dagm"]
        if cfg.lidar is not None:
            pts, hits = o["lidar"]
            d["lidar_hit"][i] = hits
            d["lidar_point"][i] = pts
    return d


INT_KEYS = ["ego_lane", "events", "done", "active", "wp_lane_index", "wp_lane_id", "wp_count", "nb_lane_index",
            "nb_lane_id", "nb_slot", "nb_count", "ogm", "dagm", "lidar_hit", "via_near", "via_near_count", "via_hit",
            "collidees", "rw_lane_count", "rw_lane", "rw_path_count", "rw_count", "rw_lane_index", "rw_lane_id"]
RW_ROWS = ["rw_pos", "rw_heading", "rw_lane_width", "rw_speed_limit", "rw_lane_index", "rw_lane_id"]


def compare(dev, ora, tol64=1e-9, tol32=2e-5, where=""):
    """Bit-exact on integer/flag arrays, tolerance on floats.  Returns list of mismatch strings."""
    bad = []
    for k, a in ora.items():
        b = dev[k]
        if k == "collidees":
            b = np.ascontiguousarray(b).view(np.uint64)  # the device tensor is int64 (torch has no uint64 arithmetic)
        if k in RW_ROWS:
            # rows beyond a path's count are not written by the device (include/smx.h): compared where counted
            R = ora["rw_count"].astype(np.int64)[..., None] > np.arange(a.shape[3])
            keep = R if a.ndim == 4 else R[..., None]
            a, b = np.where(keep, a, 0), np.where(keep, b, 0)
        if k in INT_KEYS:
            if not np.array_equal(a, b):
                idx = np.argwhere(a != b)[:4]
                bad.append(f"{where}{k}: {len(np.argwhere(a != b))} int mismatches, first {idx.tolist()} "
                           f"oracle={a[tuple(idx[0])]} dev={b[tuple(idx[0])]}")
        else:
            tol = tol32 if a.dtype == np.float32 else tol64
            with np.errstate(invalid="ignore"):
                err = np.abs(a.astype(np.float64) - b.astype(np.float64))
            err = np.where(np.isinf(a) & (a == b), 0.0, err)  # lidar misses are +inf on both sides
            err = np.where(np.isnan(err), np.inf, err)
            if a.dtype == np.float32:
                err = err / np.maximum(1.0, np.abs(a.astype(np.float64)))
            if err.size and err.max() > tol:
                idx = np.unravel_index(np.argmax(err), err.shape)
                bad.append(f"{where}{k}: max err {err.max():.3e} at {idx} oracle={a[idx]} dev={b[idx]}")
    return bad


class OracleBatch:
    """E independent oracle envs driven with the same spawns/actions as the device."""

    def __init__(self, net, cm, cfg, spawns_ep0, social_ep0=None, vias=None, missions=None):
        self.cfg = cfg
        self.road_map = ORoadNetwork(net, lanepoint_spacing=cm.lanepoint_spacing)
        self.lane_no = {lid: i for i, lid in enumerate(cm.lane_ids)}
        self.N = cfg.num_vehicles
        ocfg = oracle_config(cfg)
        K = cfg.num_social
        self.envs = []
        for e in range(cfg.num_envs):
            rows = slice(e * self.N, (e + 1) * self.N)
            social = []
            if K:
                social = [(cm.lane_ids[int(l)], float(off)) for l, off in social_ep0[rows][self.N - K:]]
            ovias = None
            if vias is not None:
                ovias = [[dict(lane_id=v.lane_id, position=v.position, hit_distance=v.hit_distance,
                               required_speed=v.required_speed) for v in lst] for lst in vias[:self.N - K]]
            omissions = None
            if missions is not None:  # smarts_amd.missions.PlannedMission | None per slot
                omissions = [dict(route=list(m.route_roads), goal=tuple(m.goal)) if m is not None else None
                             for m in missions[:self.N - K]]
            self.envs.append(OracleEnv(self.road_map, spawns_ep0[rows], [ocfg] * (self.N - K), dt=cfg.dt, social=social,
                                       social_speed_factor=cfg.social_speed_factor, vias=ovias,
                                       social_model=cfg.social_model, missions=omissions))

    def _stack(self, parts):
        return {k: np.concatenate([p[k] for p in parts], axis=0) for k in parts[0]}

    def reset_observe(self):
        return self._stack([pack(self.cfg, self.lane_no, self.N, env.reset_observe()) for env in self.envs])

    def step(self, actions):
        parts = []
        for e, env in enumerate(self.envs):
            obs, rew, dones = env.step(list(actions[e]))
            parts.append(pack(self.cfg, self.lane_no, self.N, obs, rew, dones))
        return self._stack(parts)


def sync_oracle_from_device(ob: "OracleBatch", sim):
    """Teacher forcing: overwrite every oracle vehicle's continuous state with the device's, so
    that each tick is compared as a single step from identical state (closed-loop lane following
    is bang-bang and amplifies last-ulp libm differences across ticks)."""
    import torch

    torch.cuda.synchronize()
    st = sim.state.cpu().numpy().reshape(nat.S_COUNT, -1)
    flags = sim.flags.cpu().numpy().reshape(-1)
    S = nat.S
    lane_ids = sim.cm.lane_ids
    for e, env in enumerate(ob.envs):
        for i, ag in enumerate(env.agents):
            g = e * ob.N + i
            b, c = ag.body, ag.ctrl
            b.x, b.y, b.heading = st[S["X"], g], st[S["Y"], g], st[S["HEADING"], g]
            b.u, b.v, b.yaw_rate_z, b.delta = st[S["U"], g], st[S["V"], g], st[S["R"], g], st[S["DELTA"], g]
            if hasattr(c, "integral_windup_error"):  # TrajectoryTrackingControllerState (state reuse, smx_vehicle.h)
                c.lateral_error, c.integral_velocity_error = st[S["LAT_INT"], g], st[S["SPD_INT"], g]
                c.steering_state, c.throttle_state = st[S["STEER"], g], st[S["THROTTLE"], g]
                c.velocity_error, c.integral_windup_error = st[S["SPD_ERR"], g], st[S["MCL_X"], g]
                c.heading_error = st[S["MCL_Y"], g]
            else:
                c.lateral_integral_error = st[S["LAT_INT"], g]
                c.integral_speed_error = st[S["SPD_INT"], g]
                c.steering_state = st[S["STEER"], g]
                c.throttle_state = st[S["THROTTLE"], g]
                c.speed_error = st[S["SPD_ERR"], g]
                if flags[g] & nat.F_MCL_SET:
                    c.min_curvature_location = (st[S["MCL_X"], g], st[S["MCL_Y"], g])
            ag.dist_travelled = st[S["DIST"], g]
            if ag.wps_for_distance:
                w = ag.wps_for_distance[-1]
                w.pos = np.array([st[S["TRIP_X"], g], st[S["TRIP_Y"], g]])
                w.heading = st[S["TRIP_H"], g]
            if len(ag.linear_velocities) >= 1:
                ag.linear_velocities[-1] = np.array([st[S["LV0_LONG"], g], st[S["LV0_LAT"], g], 0.0])
                ag.angular_velocities[-1] = np.array([0.0, 0.0, st[S["AV0_Z"], g]])
            if len(ag.linear_velocities) >= 2:
                ag.linear_velocities[-2] = np.array([st[S["LV1_LONG"], g], st[S["LV1_LAT"], g], 0.0])
                ag.angular_velocities[-2] = np.array([0.0, 0.0, st[S["AV1_Z"], g]])
        for k, sv in enumerate(env.social):
            g = e * ob.N + len(env.agents) + k
            b = sv.body
            b.x, b.y, b.heading, b.u = st[S["X"], g], st[S["Y"], g], st[S["HEADING"], g], st[S["U"], g]
            b.lane = ob.road_map.lane_by_id(lane_ids[int(st[S["MCL_X"], g])])
            b.offset, b.crossed = float(st[S["MCL_Y"], g]), int(st[S["SPD_INT"], g])
