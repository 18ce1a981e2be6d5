"""Batched simulation engine: E environment instances x N vehicle slots on one GPU.

Host-side owner of the device buffers (PyTorch-ROCm tensors, used purely as the
zero-copy buffer type) and thin caller of the C-ABI (include/smx.h).  It plays the
role of ``SMARTS`` (reference ``smarts/core/smarts.py``) for a whole shard:
``reset`` ~ ``SMARTS.reset`` (:365-460), ``step`` ~ ``SMARTS.step`` (:187-227).
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _native as nat
from .lidar import SensorParams, base_rays, ray_count
from .map_compiler import CompiledMap


@dataclass
class SimConfig:
    """Mirror of ``smx_config``; defaults follow the reference's AgentInterface / HiWayEnv."""

    num_envs: int = 1
    num_vehicles: int = 1
    dt: float = 0.1  # hiway_env.py:113
    waypoints: bool = True
    neighbors: bool = False
    accelerometer: bool = True  # agent_interface.py:291
    wp_lookahead: int = 32  # agent_interface.py:78
    wp_paths: int = 4  # format_obs.py:42
    wp_len: int = 20
    nb_max: int = 10  # format_obs.py:41
    nb_radius: Optional[float] = None  # agent_interface.py:98
    max_episode_steps: Optional[int] = None
    done_collision: bool = True  # agent_interface.py:186-206
    done_off_road: bool = True
    done_off_route: bool = True
    done_on_shoulder: bool = False
    done_wrong_way: bool = False
    done_not_moving: bool = False
    not_moving_time: float = 60.0
    not_moving_distance: float = 1.0
    auto_reset: bool = False  # parallel_env.py:62
    track_driven_path: bool = True
    ogm: bool = False  # agent_interface.py:42-51 (OGM defaults 256 x 256 @ 50/256)
    ogm_width: int = 256
    ogm_height: int = 256
    ogm_resolution: float = 50 / 256
    dagm: bool = False  # agent_interface.py:29-38 (DrivableAreaGridMap defaults 256 x 256 @ 50/256)
    dagm_width: int = 256
    dagm_height: int = 256
    dagm_resolution: float = 50 / 256
    lidar: Optional[SensorParams] = None  # agent_interface.py:132-135
    # DoneCriteria.agents_alive (agent_interface.py:155-176): minima (None = unset) and up to four
    # (agent slots, minimum alive) lists
    alive_min_ego: Optional[int] = None
    alive_min_total: Optional[int] = None
    alive_lists: Sequence = ()
    via_max: int = 0  # near-via rows per agent (the via sensor runs when BatchedSim gets vias)
    num_social: int = 0  # scripted social vehicles: the last num_social slots of every env (include/smx.h)
    social_speed_factor: float = 0.8
    social_model: str = "constant"  # "constant" | "idm" (include/smx.h SMX_SOCIAL_*)
    action_space: str = "Lane"  # ActionSpaceType name: Lane | Continuous | ActuatorDynamic | LaneWithContinuousSpeed
    launch_strategy: str = "auto"  # "auto" | "small" | "large": how a tick is cut into launches (include/smx.h)
    # RoadWaypoints (agent_interface.py RoadWaypoints.horizon = 32; sensors.py:991-1040): dense rows keep the first
    # rw_lanes lanes and the first rw_paths paths of each, 2 x horizon + 1 waypoints per path
    road_waypoints: bool = False
    rw_horizon: int = 32
    rw_lanes: int = 8
    rw_paths: int = 4

    def sensors_mask(self) -> int:
        m = 0
        if self.waypoints:
            m |= nat.SENSOR_WAYPOINTS
        if self.neighbors:
            m |= nat.SENSOR_NEIGHBORS
        if self.accelerometer:
            m |= nat.SENSOR_ACCELEROMETER
        if self.ogm:
            m |= nat.SENSOR_OGM
        if self.lidar is not None:
            m |= nat.SENSOR_LIDAR
        if self.dagm:
            m |= nat.SENSOR_DAGM
        if self.road_waypoints:
            m |= nat.SENSOR_ROAD_WAYPOINTS
        return m

    def done_mask(self) -> int:
        m = 0
        for bit, on in (
            (nat.DONE_COLLISION, self.done_collision), (nat.DONE_OFF_ROAD, self.done_off_road),
            (nat.DONE_OFF_ROUTE, self.done_off_route), (nat.DONE_ON_SHOULDER, self.done_on_shoulder),
            (nat.DONE_WRONG_WAY, self.done_wrong_way), (nat.DONE_NOT_MOVING, self.done_not_moving),
        ):
            if on:
                m |= bit
        return m

    def reset_elapsed_steps(self) -> int:
        """Ticks the reference burns in ``reset()`` before the first ego observation exists:
        the trap fires once ``mission.start_time`` (0.1 s, plan.py:209) has *passed*
        (trap_manager.py:53-65), and ``reset`` spins ``step({})`` until then (smarts.py:426-434)."""
        return int(math.floor(0.1 / self.dt + 1e-9)) + 1


def pack_trajectory(trajectory) -> Tuple[np.ndarray, int]:
    """(xs, ys, headings, speeds) of any length -> ([4, 11] float64, length): the first ten points and,
    in column 10, the last one — everything ``perform_trajectory_tracking_PD`` reads
    (trajectory_tracking_controller.py:176-473)."""
    n = len(trajectory[0])
    if n < 1 or any(len(r) != n for r in trajectory):
        raise ValueError("a trajectory is four equally long sequences (x, y, heading, speed)")
    out = np.zeros((4, nat.TRAJ_COLS), dtype=np.float64)
    for r in range(4):
        head = np.asarray(trajectory[r][:10], dtype=np.float64)
        out[r, :len(head)] = head
        out[r, 10] = float(trajectory[r][n - 1])
    return out, n


def lane_heading(shape: np.ndarray, seg: int) -> float:
    vx, vy = shape[seg + 1] - shape[seg]
    # heading convention of the reference (0 = +y, counter-clockwise)
    h = math.atan2(vy, vx) - math.pi / 2
    return (h + math.pi) % (2 * math.pi) - math.pi


def make_spawns(cm: CompiledMap, num_envs: int, num_vehicles: int, episodes: int = 1, seed: int = 42,
                first_env: int = 0, min_gap: float = 8.0, return_lanes: bool = False):
    """Synthetic spawn table: vehicle k of env e starts on lane (k mod L) of the
    map's normal lanes at an arclength drawn U(0.05, 0.95)*lane_length from
    ``numpy.random.Generator(PCG64(seed + e))``, re-drawn while within ``min_gap`` metres of an
    earlier vehicle on that lane; heading = lane heading, speed = lane speed limit.
    Returns ``[episodes, num_envs * num_vehicles, 4]`` (x, y, heading, speed); with
    ``return_lanes`` also ``[episodes, num_envs * num_vehicles, 2]`` (lane index, arclength), which
    is what scripted social vehicles start from."""
    lanes = [i for i in range(cm.n_lanes) if not cm.lane_in_junction[i]]
    shapes = [cm.lane_shape(i) for i in lanes]
    cums = []
    for sh in shapes:
        acc, cum = 0.0, [0.0]
        for a, b in zip(sh[:-1], sh[1:]):  # vertex by vertex, like smx_shape_rec.cum
            ex, ey = float(a[0] - b[0]), float(a[1] - b[1])
            acc = acc + math.sqrt(ex * ex + ey * ey)
            cum.append(acc)
        cums.append(np.array(cum))
    out = np.zeros((episodes, num_envs * num_vehicles, 4), dtype=np.float64)
    where = np.zeros((episodes, num_envs * num_vehicles, 2), dtype=np.float64)
    L = len(lanes)
    for e in range(num_envs):
        rng = np.random.Generator(np.random.PCG64(seed + first_env + e))
        for ep in range(episodes):
            taken: Dict[int, list] = {}
            for k in range(num_vehicles):
                li = k % L
                cum = cums[li]
                total = cum[-1]
                for _ in range(1000):
                    off = rng.uniform(0.05, 0.95) * total
                    if all(abs(off - o) >= min_gap for o in taken.get(li, ())):
                        break
                taken.setdefault(li, []).append(off)
                seg = int(np.searchsorted(cum, off, side="right") - 1)
                seg = min(max(seg, 0), len(cum) - 2)
                f = (off - cum[seg]) / (cum[seg + 1] - cum[seg]) if cum[seg + 1] > cum[seg] else 0.0
                sh = shapes[li]
                x = sh[seg, 0] + (sh[seg + 1, 0] - sh[seg, 0]) * f
                y = sh[seg, 1] + (sh[seg + 1, 1] - sh[seg, 1]) * f
                out[ep, e * num_vehicles + k] = (x, y, lane_heading(sh, seg), cm.lane_speed[lanes[li]])
                where[ep, e * num_vehicles + k] = (lanes[li], off)
    return (out, where) if return_lanes else out


from .map_compiler import map_tables_struct  # noqa: E402,F401  (torch-free; also used by tests/native)


class BatchedSim:
    """One shard of environment instances resident on one GPU."""

    def __init__(self, cm: CompiledMap, cfg: SimConfig, device: str = "cuda:0", spawns: Optional[np.ndarray] = None,
                 spawn_episodes: int = 2, seed: int = 42, first_env: int = 0, social_spawns: Optional[np.ndarray] = None,
                 vias: Optional[Sequence[Sequence]] = None, missions: Optional[Sequence] = None):
        """``vias``: per agent slot, a list of ``vias.ResolvedVia`` (the missions' via points).
        ``missions``: per vehicle slot ``None`` (endless mission, empty route) or a
        ``missions.PlannedMission`` (fixed route + PositionalGoal); see ``set_missions``."""
        self.lib = nat.load_library()
        if not torch.cuda.is_available():
            raise nat.NativeLibraryError("no ROCm device visible: the smarts_amd hot path runs on the GPU only")
        self.cm = cm
        self.cfg = cfg
        self.device = torch.device(device)
        E, N = cfg.num_envs, cfg.num_vehicles
        self.E, self.N = E, N
        dev = self.device
        idx = self.device.index if self.device.index is not None else 0

        c = nat.SmxConfig()
        c.num_envs, c.num_vehicles, c.dt = E, N, cfg.dt
        c.sensors, c.done_criteria = cfg.sensors_mask(), cfg.done_mask()
        c.wp_lookahead, c.wp_paths, c.wp_len = cfg.wp_lookahead, cfg.wp_paths, cfg.wp_len
        c.nb_max = cfg.nb_max
        c.nb_radius = -1.0 if cfg.nb_radius is None else float(cfg.nb_radius)
        c.max_episode_steps = cfg.max_episode_steps or 0
        c.not_moving_time, c.not_moving_distance = cfg.not_moving_time, cfg.not_moving_distance
        c.auto_reset = 1 if cfg.auto_reset else 0
        c.reset_elapsed_steps = cfg.reset_elapsed_steps()
        if cfg.action_space not in nat.ACTION_SPACES:
            raise ValueError(f"action space {cfg.action_space!r} is not on the accelerated path "
                             f"(supported: {sorted(nat.ACTION_SPACES)})")
        c.action_space = nat.ACTION_SPACES[cfg.action_space]
        c.num_social, c.social_speed_factor = int(cfg.num_social), float(cfg.social_speed_factor)
        c.social_model = nat.SOCIAL_MODELS[cfg.social_model]
        self.vias = [list(v) for v in vias] if vias is not None else None
        if self.vias is not None and cfg.via_max <= 0:
            raise ValueError("vias need SimConfig(via_max > 0)")
        c.via_max = int(cfg.via_max)
        c.alive_min_ego, c.alive_min_total = int(cfg.alive_min_ego or 0), int(cfg.alive_min_total or 0)
        if len(cfg.alive_lists) > 4:
            raise ValueError("DoneCriteria.agents_alive: at most four agent lists on the accelerated path")
        c.alive_lists = len(cfg.alive_lists)
        for k, (slots, minimum) in enumerate(cfg.alive_lists):
            mask = 0
            for i in slots:
                if not 0 <= int(i) < N - cfg.num_social:
                    raise ValueError(f"agents_alive list {k}: slot {i} is not an agent slot")
                mask |= 1 << int(i)
            c.alive_list_mask[k], c.alive_list_min[k] = mask, int(minimum)
        if cfg.ogm:
            c.ogm_width, c.ogm_height, c.ogm_resolution = cfg.ogm_width, cfg.ogm_height, cfg.ogm_resolution
        if cfg.lidar is not None:
            c.lidar_rays, c.lidar_max_distance = ray_count(cfg.lidar), cfg.lidar.max_distance
        if cfg.dagm:
            c.dagm_width, c.dagm_height, c.dagm_resolution = cfg.dagm_width, cfg.dagm_height, cfg.dagm_resolution
        if cfg.road_waypoints:
            c.rw_horizon, c.rw_lanes, c.rw_paths = int(cfg.rw_horizon), int(cfg.rw_lanes), int(cfg.rw_paths)
        self._c = c
        self.handle = C.c_void_p()
        rc = self.lib.smx_create(C.byref(c), idx, C.byref(self.handle))
        nat.check(self.lib, self.handle, rc, "smx_create")
        rc = self.lib.smx_set_launch_strategy(self.handle, nat.LAUNCH_STRATEGIES[cfg.launch_strategy])
        nat.check(self.lib, self.handle, rc, "smx_set_launch_strategy")
        tables, keep = map_tables_struct(cm)
        nat.check(self.lib, self.handle, self.lib.smx_load_map(self.handle, C.byref(tables)), "smx_load_map")
        del keep
        if self.vias is not None:
            if len(self.vias) != N:
                raise ValueError("vias: one list per vehicle slot")
            flat = [v for lst in self.vias for v in lst]
            recs = (nat.SmxVia * max(len(flat), 1))()
            for i, v in enumerate(flat):
                recs[i].x, recs[i].y = v.position
                recs[i].hit_distance, recs[i].required_speed, recs[i].lane = v.hit_distance, v.required_speed, v.lane
            offs = (C.c_int32 * (N + 1))(*np.concatenate([[0], np.cumsum([len(lst) for lst in self.vias])]).astype(int).tolist())
            nat.check(self.lib, self.handle, self.lib.smx_set_vias(self.handle, recs, len(flat), offs), "smx_set_vias")
        self.missions = None
        if missions is not None:
            self.set_missions(missions)
        if cfg.lidar is not None:
            self.lidar_rays = torch.from_numpy(base_rays(cfg.lidar)).to(dev)
            rc = self.lib.smx_set_lidar_rays(self.handle, self.lidar_rays.data_ptr(), int(self.lidar_rays.shape[0]))
            nat.check(self.lib, self.handle, rc, "smx_set_lidar_rays")

        # ---- state ----
        T = E * N
        self.state = torch.zeros((nat.S_COUNT, E, N), dtype=torch.float64, device=dev)
        self.flags = torch.zeros((E, N), dtype=torch.int32, device=dev)
        self.steps = torch.zeros((E, N), dtype=torch.int32, device=dev)
        self.env_ticks = torch.zeros((E,), dtype=torch.int32, device=dev)
        self.env_done_count = torch.zeros((E,), dtype=torch.int32, device=dev)
        self.env_episode = torch.full((E,), -1, dtype=torch.int32, device=dev)
        need_ring = cfg.track_driven_path or cfg.done_not_moving
        self.driven_path = torch.zeros((T, nat.DRIVEN_PATH_LEN), dtype=torch.float64, device=dev) if need_ring else None
        self.seed_cache = torch.full((nat.SEED_COUNT, E, N), -1, dtype=torch.int32, device=dev)
        self.facts_i32 = torch.full((nat.FACT_I_COUNT, E, N), -1, dtype=torch.int32, device=dev)
        self.facts_f64 = torch.zeros((nat.FACT_F_COUNT, E, N), dtype=torch.float64, device=dev)
        self.env_reset_pending = torch.zeros((E,), dtype=torch.int32, device=dev)
        st = nat.SmxState()
        for name, t in (("f64", self.state), ("flags", self.flags), ("steps", self.steps), ("env_ticks", self.env_ticks),
                        ("env_done_count", self.env_done_count), ("env_episode", self.env_episode),
                        ("driven_path", self.driven_path), ("seed_cache", self.seed_cache),
                        ("facts_i32", self.facts_i32), ("facts_f64", self.facts_f64),
                        ("env_reset_pending", self.env_reset_pending)):
            nat.bind_buffer(st, nat.STATE_BUFFERS, name, t)  # pointer + element count + dtype (checked on entry)
        self._st = st

        # ---- spawns ----
        if spawns is None:
            spawns, where = make_spawns(cm, E, N, episodes=spawn_episodes, seed=seed, first_env=first_env,
                                        return_lanes=True)
            if social_spawns is None:
                social_spawns = where
        spawns = np.ascontiguousarray(spawns, dtype=np.float64)
        assert spawns.ndim == 3 and spawns.shape[1:] == (T, 4), spawns.shape
        self.spawns = torch.from_numpy(spawns).to(dev)
        sp = nat.SmxSpawns()
        sp.episodes, sp.pose, sp.pose_count = int(spawns.shape[0]), self.spawns.data_ptr(), self.spawns.numel()
        self.social_spawns = None
        if cfg.num_social > 0:
            if social_spawns is None:
                raise ValueError("num_social > 0 with an explicit spawn table needs social_spawns "
                                 "(make_spawns(..., return_lanes=True))")
            social_spawns = np.ascontiguousarray(social_spawns, dtype=np.float64)
            assert social_spawns.shape == (spawns.shape[0], T, 2), social_spawns.shape
            self.social_spawns = torch.from_numpy(social_spawns).to(dev)
            sp.social, sp.social_count = self.social_spawns.data_ptr(), self.social_spawns.numel()
        self._sp = sp

        # ---- outputs (dense StdObs layout, format_obs.py:313-373) ----
        o: Dict[str, torch.Tensor] = {}
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)  # noqa: E731
        o["ego_pos"] = z((E, N, 3), torch.float64)
        o["ego_f32"] = z((E, N, nat.EGO_F32_COUNT), torch.float32)
        o["ego_lane"] = z((E, N, 2), torch.int16)
        o["events"] = z((E, N, nat.EV_COUNT), torch.uint8)
        o["reward"] = z((E, N), torch.float64)
        o["dist"] = z((E, N), torch.float64)
        o["done"] = z((E, N), torch.uint8)
        o["active"] = z((E, N), torch.uint8)
        o["env_done"] = z((E,), torch.uint8)
        if cfg.waypoints:
            P, W = cfg.wp_paths, cfg.wp_len
            o["wp_pos"] = z((E, N, P, W, 3), torch.float64)
            o["wp_heading"] = z((E, N, P, W), torch.float32)
            o["wp_lane_width"] = z((E, N, P, W), torch.float32)
            o["wp_speed_limit"] = z((E, N, P, W), torch.float32)
            o["wp_lane_index"] = z((E, N, P, W), torch.int8)
            o["wp_lane_id"] = z((E, N, P, W), torch.int16)
            o["wp_count"] = z((E, N, P + 1), torch.uint8)
        if cfg.neighbors:
            K = cfg.nb_max
            o["nb_pos"] = z((E, N, K, 3), torch.float64)
            o["nb_box"] = z((E, N, K, 3), torch.float32)
            o["nb_heading"] = z((E, N, K), torch.float32)
            o["nb_speed"] = z((E, N, K), torch.float32)
            o["nb_lane_index"] = z((E, N, K), torch.int8)
            o["nb_lane_id"] = z((E, N, K), torch.int16)
            o["nb_slot"] = z((E, N, K), torch.int8)
            o["nb_count"] = z((E, N), torch.uint8)
        if cfg.via_max > 0:
            o["via_near"] = torch.full((E, N, cfg.via_max), -1, dtype=torch.int8, device=dev)
            o["via_near_count"] = z((E, N), torch.uint8)
            o["via_hit"] = z((E, N), torch.int32)
        if cfg.ogm:
            o["ogm"] = z((E, N, cfg.ogm_height, cfg.ogm_width), torch.uint8)
        if cfg.dagm:
            o["dagm"] = z((E, N, cfg.dagm_height, cfg.dagm_width), torch.uint8)
        if cfg.road_waypoints:
            L, Q, R = cfg.rw_lanes, cfg.rw_paths, 2 * cfg.rw_horizon + 1
            o["rw_lane_count"] = z((E, N), torch.uint8)
            o["rw_lane"] = torch.full((E, N, L), -1, dtype=torch.int16, device=dev)
            o["rw_path_count"] = z((E, N, L), torch.int16)
            o["rw_count"] = z((E, N, L, Q), torch.uint8)
            o["rw_pos"] = z((E, N, L, Q, R, 3), torch.float64)
            o["rw_heading"] = z((E, N, L, Q, R), torch.float32)
            o["rw_lane_width"] = z((E, N, L, Q, R), torch.float32)
            o["rw_speed_limit"] = z((E, N, L, Q, R), torch.float32)
            o["rw_lane_index"] = z((E, N, L, Q, R), torch.int8)
            o["rw_lane_id"] = z((E, N, L, Q, R), torch.int16)
        if cfg.lidar is not None:
            R = ray_count(cfg.lidar)
            o["lidar_hit"] = z((E, N, R), torch.uint8)
            o["lidar_point"] = z((E, N, R, 3), torch.float64)
        # learner-facing block (reward, done) as float32, two buffers used on alternate ticks so that
        # one can be in flight in a collective while the next tick writes the other
        self._learner = [z((2, E, N), torch.float32), z((2, E, N), torch.float32)]
        self._learner_k = 0
        o["learner"] = self._learner[0]
        self.out = o
        o["collidees"] = z((E, N), torch.int64)  # bit j: touching the vehicle in slot j (read as unsigned)
        if cfg.auto_reset:
            # the finishing tick's low-dimensional rows of an env that restarts inside the launch (include/smx.h)
            o["final_ego_pos"] = z((E, N, 3), torch.float64)
            o["final_ego_f32"] = z((E, N, nat.EGO_F32_COUNT), torch.float32)
            o["final_ego_lane"] = z((E, N, 2), torch.int16)
            o["final_events"] = z((E, N, nat.EV_COUNT), torch.uint8)
            o["final_dist"] = z((E, N), torch.float64)
        so = nat.SmxOutputs()
        for name in nat.OUTPUT_FIELDS:
            nat.bind_buffer(so, nat.OUTPUT_FIELDS, name, o.get(name))
        self._out = so
        self._stream = None
        self._was_reset = False

    # ------------------------------------------------------------------
    def _stream_ptr(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def output_bytes_per_agent_step(self) -> int:
        """Bytes of observation/reward/done written per agent-step (dense layout)."""
        per = 0
        for name, t in self.out.items():
            if name in ("env_done", "learner") or name.startswith("final_"):
                continue
            per += t[0, 0].numel() * t.element_size()
        return per

    def state_bytes_per_agent_step(self) -> int:
        return nat.S_COUNT * 8 + 4 + 4 + nat.SEED_COUNT * 4

    def kernel_bytes_per_agent_step(self) -> Dict[str, Tuple[int, int]]:
        """Algorithmic HBM bytes of one agent-step as (read, write) per timing phase — SURVEY.md §8(d)'s
        accounting on this layout: compulsory reads of the vehicle's own state and action, the state written
        back, and every observation / reward / done byte of the dense rows.  NOT counted: the map tables
        (L2 / Infinity-Cache resident) and the hand-off buffers between the kernels of a tick (path seeds,
        road facts: ``handoff_bytes_per_agent_step``) — traffic this design adds, not traffic the path needs."""
        o = {k: (t[0, 0].numel() * t.element_size()) for k, t in self.out.items()
             if k not in ("env_done", "learner") and not k.startswith("final_")}  # (final_*: copied for restarting envs only)
        pose = 3 * 8 + 4  # x, y, heading + flags: what every sensor kernel reads of a vehicle
        ctrl_state = 14 * 8 + 4  # SMX_S_X .. SMX_S_MCL_Y + flags: read and written back by k_control
        obs_state_r, obs_state_w = 16 * 8 + 4 + 4, 12 * 8 + 4  # trip meter / accelerometer / driven-path fields, steps
        wp = sum(v for k, v in o.items() if k.startswith("wp_")) if self.cfg.waypoints else 0
        rows = sum(v for k, v in o.items() if not k.startswith(("wp_", "ogm", "lidar", "dagm", "rw_"))) + 8  # + learner block
        kb = {
            "control": (ctrl_state + (12 if self.cfg.action_space != "Lane" else 1), ctrl_state + 2 * 8),
            "scan": (pose, 0),
            "sensors": ((pose if self.cfg.waypoints else 0) + obs_state_r, wp + rows + obs_state_w),
            "commit": (4, 4),
        }
        ogm = (pose, o["ogm"]) if self.cfg.ogm else (0, 0)
        dagm = (pose, o["dagm"]) if self.cfg.dagm else (0, 0)
        tile = self.cfg.ogm_width * self.cfg.ogm_height
        ogm_env_small = (self.small_form() and self.E * self.N >= nat.OGM_ENV_MIN_VEHICLES and self.N <= 32
                         and tile * 8 <= 64 * 1024)  # smx_kernels.hip enqueue(): k_ogm_env on small batches too
        ogm_inline = (self.cfg.ogm and tile <= 16 * 1024
                      and self.small_form() and not ogm_env_small)  # smx_kernels.hip enqueue(): small batches only
        add = lambda a, b: (a[0] + b[0], a[1] + b[1])  # noqa: E731
        if (self.cfg.ogm and not ogm_inline) or self.cfg.dagm:
            kb["ogm"] = add((0, 0) if ogm_inline else ogm, dagm)  # their own launches (one timing phase)
        if ogm_inline:
            kb["sensors"] = add(kb["sensors"], ogm)
        if self.cfg.lidar is not None:
            kb["sensors"] = add(kb["sensors"], (pose, o["lidar_hit"] + o["lidar_point"]))
        return kb

    def set_missions(self, missions: Optional[Sequence]):
        """Fixed-route missions per vehicle slot, shared by every env (``smx_set_missions``): ``None`` entries
        (or ``missions=None``) are endless missions with an empty route."""
        N = self.N
        if missions is None or all(m is None for m in missions):
            nat.check(self.lib, self.handle, self.lib.smx_set_missions(self.handle, None, 0, None, 0), "smx_set_missions")
            self.missions = None
            return
        if len(missions) != N:
            raise ValueError("missions: one entry per vehicle slot")
        road_no = {rid: i for i, rid in enumerate(self.cm.road_ids)}
        recs = (nat.SmxMission * N)()
        roads = []
        for s, m in enumerate(missions):
            if m is None:
                continue
            recs[s].goal_x, recs[s].goal_y, recs[s].goal_radius = m.goal
            recs[s].route_off, recs[s].route_len = len(roads), len(m.route_roads)
            roads += [road_no[r] for r in m.route_roads]
        arr = (C.c_int32 * max(len(roads), 1))(*roads)
        nat.check(self.lib, self.handle, self.lib.smx_set_missions(self.handle, recs, N, arr, len(roads)), "smx_set_missions")
        self.missions = list(missions)

    def small_form(self) -> bool:
        """Whether a tick runs in the SMALL launch form (smx_kernels.hip: SMX_LARGE_BATCH_VEHICLES)."""
        s = self.cfg.launch_strategy
        return s == "small" or (s == "auto" and self.E * self.N <= nat.LARGE_BATCH_VEHICLES)  # ("large*": never)

    def launch_form(self) -> str:
        """The form the library runs a tick in: "small", "large_teams" or "large_one_lane" (smx_launch_form)."""
        rc = self.lib.smx_launch_form(self.handle)
        if rc < 0:
            nat.check(self.lib, self.handle, rc, "smx_launch_form")
        return nat.LAUNCH_FORMS[rc]

    def handoff_bytes_per_agent_step(self) -> int:
        """Bytes written by one kernel of the tick and read by a later one (path seeds, road facts, next flags):
        reported beside the algorithmic bytes, never inside them."""
        return 2 * (nat.SEED_COUNT * 4 + nat.FACT_I_COUNT * 4 + nat.FACT_F_COUNT * 8) + nat.SEED_COUNT * 4

    def reset(self, env_mask: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        mask_ptr = None
        if env_mask is not None:
            env_mask = env_mask.to(device=self.device, dtype=torch.uint8).contiguous()
            assert env_mask.shape == (self.E,)
            mask_ptr = env_mask.data_ptr()
        rc = self.lib.smx_reset(self.handle, mask_ptr, C.byref(self._st), C.byref(self._sp), C.byref(self._out),
                                self._stream_ptr())
        nat.check(self.lib, self.handle, rc, "smx_reset")
        self._was_reset = True
        return self.out

    def step(self, actions: torch.Tensor) -> Dict[str, torch.Tensor]:
        if not self._was_reset:
            raise RuntimeError("step() before reset()")  # SMARTSNotSetupError (smarts.py:207-208)
        if self.cfg.action_space == "Trajectory":
            raise ValueError("ActionSpaceType.Trajectory steps through step_trajectory(trajectories, counts)")
        lane = self.cfg.action_space == "Lane"
        want_dtype, want_shape = (torch.int8, (self.E, self.N)) if lane else (torch.float32, (self.E, self.N, 3))
        if actions.dtype != want_dtype or actions.device != self.device or not actions.is_contiguous():
            actions = actions.to(device=self.device, dtype=want_dtype).contiguous()
        if tuple(actions.shape) != want_shape:
            raise ValueError(f"{self.cfg.action_space} actions must have shape {want_shape}, got {tuple(actions.shape)}")
        # only now that the call is known to go ahead does the learner block flip
        self._learner_k ^= 1
        self.out["learner"] = self._learner[self._learner_k]
        self._out.learner = self.out["learner"].data_ptr()  # same extent and dtype as the other block
        if lane:
            # (codes outside -1..3 cannot be seen from here without a device round trip: the kernel treats them
            # as "no action" and smx_sync / BatchedSim.sync() reports them)
            rc = self.lib.smx_step(self.handle, actions.data_ptr(), C.byref(self._st), C.byref(self._sp),
                                   C.byref(self._out), self._stream_ptr())
            nat.check(self.lib, self.handle, rc, "smx_step")
        else:
            # three floats per agent; NaN in the first one = no action this tick
            rc = self.lib.smx_step_continuous(self.handle, actions.data_ptr(), C.byref(self._st), C.byref(self._sp),
                                              C.byref(self._out), self._stream_ptr())
            nat.check(self.lib, self.handle, rc, "smx_step_continuous")
        return self.out

    @property
    def next_learner_block(self) -> torch.Tensor:
        """The learner block the NEXT ``step`` will write (see ``RewardDoneGather.release``)."""
        return self._learner[self._learner_k ^ 1]

    def step_trajectory(self, trajectories: torch.Tensor, counts: torch.Tensor) -> Dict[str, torch.Tensor]:
        """One tick in ActionSpaceType.Trajectory: ``trajectories`` float64 [E, N, 4, 11] in the packed
        form of include/smx.h (``pack_trajectory``), ``counts`` int32 [E, N] (0 = no action)."""
        if not self._was_reset:
            raise RuntimeError("step() before reset()")
        if self.cfg.action_space != "Trajectory":
            raise ValueError("step_trajectory needs SimConfig(action_space='Trajectory')")
        trajectories = trajectories.to(device=self.device, dtype=torch.float64).contiguous()
        counts = counts.to(device=self.device, dtype=torch.int32).contiguous()
        assert trajectories.shape == (self.E, self.N, 4, nat.TRAJ_COLS) and counts.shape == (self.E, self.N)
        self._learner_k ^= 1
        self.out["learner"] = self._learner[self._learner_k]
        self._out.learner = self.out["learner"].data_ptr()
        rc = self.lib.smx_step_trajectory(self.handle, trajectories.data_ptr(), counts.data_ptr(), C.byref(self._st),
                                          C.byref(self._sp), C.byref(self._out), self._stream_ptr())
        nat.check(self.lib, self.handle, rc, "smx_step_trajectory")
        return self.out

    def set_timing(self, level):
        """0/False = off, 1/True = one event pair per smx_step, 2 = per-kernel phases."""
        nat.check(self.lib, self.handle, self.lib.smx_set_timing(self.handle, int(level)), "smx_set_timing")

    def read_phase_ms(self, max_steps: int = 16384) -> np.ndarray:
        """[steps, len(PHASES)] kernel-phase durations (ms) recorded at timing level 2."""
        k = len(nat.PHASES)
        buf = (C.c_float * (max_steps * k))()
        n = C.c_int32()
        rc = self.lib.smx_read_phase_ms(self.handle, buf, max_steps, C.byref(n))
        nat.check(self.lib, self.handle, rc, "smx_read_phase_ms")
        return np.frombuffer(buf, dtype=np.float32, count=n.value * k).reshape(n.value, k).copy()

    def last_step_ms(self) -> float:
        ms = C.c_float()
        nat.check(self.lib, self.handle, self.lib.smx_last_step_ms(self.handle, C.byref(ms)), "smx_last_step_ms")
        return float(ms.value)

    def read_step_ms(self, max_count: int = 65536) -> np.ndarray:
        """Durations (ms) of the smx_step launches recorded since timing was enabled / last read."""
        buf = (C.c_float * max_count)()
        n = C.c_int32()
        rc = self.lib.smx_read_step_ms(self.handle, buf, max_count, C.byref(n))
        nat.check(self.lib, self.handle, rc, "smx_read_step_ms")
        return np.frombuffer(buf, dtype=np.float32, count=n.value).copy()

    def sync(self):
        nat.check(self.lib, self.handle, self.lib.smx_sync(self.handle, self._stream_ptr()), "smx_sync")

    def close(self):
        if getattr(self, "handle", None):
            self.lib.smx_destroy(self.handle)
            self.handle = None

    def __del__(self):  # smarts.py:711-721
        try:
            self.close()
        except Exception:
            pass
