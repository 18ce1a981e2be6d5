"""Shared engine behind ``HiWayEnv`` and ``ParallelEnv``: E identical env instances x N agents on
one GPU (``BatchedSim``), with the reference's dict-of-agent-id surface on top.

Plays the role of ``SMARTS`` + ``AgentManager`` (reference ``smarts/core/smarts.py:187-227, 365-460``,
``agent_manager.py:161-240``) for the whole batch.
"""
from __future__ import annotations

import os
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .. import _native as nat
from ..lidar import base_rays
from .agent import AgentSpec
from .agent_interface import ActionSpaceType, AgentInterface
from .observations import Observation, ObservationBuilder

# Controllers.perform_action, Lane space (controllers/__init__.py:125-144)
LANE_ACTIONS = {"keep_lane": 0, "slow_down": 1, "change_lane_left": 2, "change_lane_right": 3}
NO_ACTION = -1

PKG_SCENARIOS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenarios")


class SMARTSNotSetupError(Exception):
    """smarts.py:73-76: stepping before reset."""


class SMARTSDestroyedError(Exception):
    """smarts.py:79-82: use after destroy."""


def resolve_scenario(path: str) -> str:
    """A scenario directory holding ``map.net.xml`` (the reference's layout, scenario.py:109-121) or
    the compact ``map.smxnet.json.gz``; a bare reference-style name (``scenarios/loop``) falls back
    to the maps shipped with the package."""
    cands = [path]
    parts = os.path.normpath(path).split(os.sep)
    if "scenarios" in parts:
        tail = parts[parts.index("scenarios") + 1:]
        cands.append(os.path.join(PKG_SCENARIOS, *tail))
    cands.append(os.path.join(PKG_SCENARIOS, os.path.basename(os.path.normpath(path))))
    for c in cands:
        if os.path.isdir(c) and any(os.path.exists(os.path.join(c, f)) for f in ("map.net.xml", "map.smxnet.json.gz")):
            return c
    raise FileNotFoundError(f"scenario {path!r}: no map.net.xml / map.smxnet.json.gz found (looked in {cands})")


def encode_lane_action(action: Any) -> int:
    """Lane action string -> device code; anything else is an error, as in the reference
    (controllers/__init__.py:137-144 looks the string up in a dict)."""
    if isinstance(action, str):
        if action not in LANE_ACTIONS:
            raise KeyError(f"unknown Lane action {action!r}; expected one of {sorted(LANE_ACTIONS)}")
        return LANE_ACTIONS[action]
    raise TypeError(f"ActionSpaceType.Lane expects a string action, got {type(action).__name__}")


def encode_float_action(space: ActionSpaceType, action: Any):
    """Continuous / ActuatorDynamic: (throttle, brake, steering[-rate]); LaneWithContinuousSpeed:
    (target_speed, lane_change) (controllers/__init__.py:94-124) -> three float32."""
    vals = [float(x) for x in action]
    if space is ActionSpaceType.LaneWithContinuousSpeed:
        if len(vals) != 2:
            raise ValueError("LaneWithContinuousSpeed expects (target_speed, lane_change)")
        vals.append(0.0)
    elif len(vals) != 3:
        raise ValueError(f"{space.name} expects three floats")
    return vals


# paths kept per agent when the caller asks for the reference's full ``waypoint_paths`` (one path per
# lane of the ego's road, plus junction branches; the shipped maps have at most 4 lanes per road)
FULL_WINDOW_PATHS = 8


def sim_config_from_interface(itf: AgentInterface, num_envs: int, num_agents: int, dt: float, auto_reset: bool,
                              waypoint_window: Optional[Tuple[int, int]] = (4, 20), num_social: int = 0,
                              agent_ids: Optional[Sequence[str]] = None):
    """AgentInterface -> SimConfig (one interface for every agent, as FormatObs also requires,
    format_obs.py:207-210).  ``waypoint_window`` = (paths, waypoints per path) kept in the dense
    rows; ``None`` keeps what the reference's ``Observation`` holds: every waypoint of a path
    (``lookahead + 1``) for up to ``FULL_WINDOW_PATHS`` paths."""
    from ..engine import SimConfig

    itf.validate_for_device()
    dc, evc = itf.done_criteria, itf.event_configuration
    kw: Dict[str, Any] = dict(
        num_envs=num_envs, num_vehicles=num_agents + num_social, num_social=num_social, dt=dt,
        waypoints=bool(itf.waypoints), wp_lookahead=itf.waypoints.lookahead if itf.waypoints else 32,
        neighbors=bool(itf.neighborhood_vehicles),
        nb_radius=itf.neighborhood_vehicles.radius if itf.neighborhood_vehicles else None,
        accelerometer=bool(itf.accelerometer), max_episode_steps=itf.max_episode_steps,
        done_collision=dc.collision, done_off_road=dc.off_road, done_off_route=dc.off_route,
        done_on_shoulder=dc.on_shoulder, done_wrong_way=dc.wrong_way, done_not_moving=dc.not_moving,
        not_moving_time=evc.not_moving_time, not_moving_distance=evc.not_moving_distance, auto_reset=auto_reset,
        action_space=itf.action.name if itf.action is not None else "Lane",  # None: every action is "no action"
    )
    alive = dc.agents_alive
    if alive is not None:
        kw.update(alive_min_ego=alive.minimum_ego_agents_alive, alive_min_total=alive.minimum_total_agents_alive)
        lists = []
        for lst in alive.agent_lists_alive or ():
            if agent_ids is None:
                raise ValueError("agents_alive lists need the agent ids")
            # ids that are not agents of this env are never alive (sensors.py:432-435)
            lists.append(([agent_ids.index(a) for a in lst.agents_list if a in agent_ids], lst.minimum_agents_alive_in_list))
        kw["alive_lists"] = tuple(lists)
    if itf.waypoints:
        # (4, 20) is the StdObs window (format_obs.py:42); rows are never longer than the lookahead
        if waypoint_window is None:
            waypoint_window = (FULL_WINDOW_PATHS, itf.waypoints.lookahead + 1)
        kw["wp_paths"] = waypoint_window[0]
        kw["wp_len"] = min(waypoint_window[1], itf.waypoints.lookahead + 1)
    if itf.ogm:
        kw.update(ogm=True, ogm_width=itf.ogm.width, ogm_height=itf.ogm.height, ogm_resolution=itf.ogm.resolution)
    if itf.drivable_area_grid_map:
        g = itf.drivable_area_grid_map
        kw.update(dagm=True, dagm_width=g.width, dagm_height=g.height, dagm_resolution=g.resolution)
    if itf.lidar:
        kw.update(lidar=itf.lidar.sensor_params)
    if itf.road_waypoints:
        kw.update(road_waypoints=True, rw_horizon=itf.road_waypoints.horizon)
    return SimConfig(**kw)


class BatchCore:
    """E env instances of one scenario x the agents of ``agent_specs`` on one device."""

    def __init__(self, scenario_dir: str, agent_specs: Dict[str, AgentSpec], num_envs: int, dt: float, seed: int,
                 auto_reset: bool, device: str = "cuda:0", waypoint_window: Optional[Tuple[int, int]] = (4, 20),
                 num_social: int = 0, vias: Optional[Dict[str, Sequence]] = None, social_model: str = "constant",
                 missions: Optional[Dict[str, Any]] = None, spawns: str = "reference", shuffle_scenarios: bool = True):
        from ..engine import BatchedSim, make_spawns
        from ..scenario_build import load_compiled_map

        self.agent_ids: List[str] = list(agent_specs.keys())
        self.agent_specs = agent_specs
        interfaces = [spec.interface for spec in agent_specs.values()]
        if any(i is None for i in interfaces):
            raise ValueError("every AgentSpec needs an interface")
        first = interfaces[0]
        if any(i != first for i in interfaces[1:]):
            raise NotImplementedError("all agents of an accelerated env must share one AgentInterface")
        self.interface: AgentInterface = first
        self.E, self.N, self.dt, self.seed = num_envs, len(self.agent_ids), dt, seed
        self.scenario_dir = resolve_scenario(scenario_dir)
        self.cm = load_compiled_map(self.scenario_dir)  # compiled-map cache next to the map (scenario build)
        self.cfg = sim_config_from_interface(first, num_envs, self.N, dt, auto_reset, waypoint_window, num_social,
                                             self.agent_ids)
        self.num_social = num_social
        self.cfg.social_model = social_model
        # Start poses.  "reference": what hiway-v0 gives agents of a scenario without missions.pkl — a random endless
        # mission each, from CPython's random stream (missions.reference_spawn_table); "synthetic": the benchmark's
        # spawn table (PCG64(seed + env), 8 m apart on a lane, at the speed limit).  Scripted social
        # vehicles (the stand-in for the scenario's SUMO flows) always start from the synthetic table.
        if spawns not in ("reference", "synthetic"):
            raise ValueError('spawns must be "reference" or "synthetic"')
        spawn_mode = spawns
        spawns, where = make_spawns(self.cm, num_envs, self.N + num_social, episodes=4, seed=seed, return_lanes=True)
        if spawn_mode == "reference":
            from ..missions import reference_spawn_table
            from ..sumo_map import load_net

            ref = reference_spawn_table(load_net(self.scenario_dir), num_envs, self.N, seed, episodes=spawns.shape[0],
                                        shuffle_scenarios=shuffle_scenarios)
            slots = self.N + num_social
            spawns.reshape(spawns.shape[0], num_envs, slots, 4)[:, :, :self.N] = ref.reshape(spawns.shape[0], num_envs, self.N, 4)
        # mission vias (sstudio Via per agent id) -> resolved lists per vehicle slot
        self.vias = None
        if vias:
            from ..vias import resolve_vias

            unknown = set(vias) - set(self.agent_ids)
            if unknown:
                raise ValueError(f"vias for unknown agents: {sorted(unknown)}")
            self.vias = [resolve_vias(self.cm, vias.get(a, ())) for a in self.agent_ids] + [[] for _ in range(num_social)]
            self.cfg.via_max = 8
        # fixed-route missions (sstudio Mission per agent id): planned once (Scenario._extract_mission +
        # Plan.create_route), the agent's spawn rows become the mission's start in every env and episode
        self.missions = None
        if missions:
            from ..missions import plan_mission
            from ..sumo_map import load_net

            unknown = set(missions) - set(self.agent_ids)
            if unknown:
                raise ValueError(f"missions for unknown agents: {sorted(unknown)}")
            net = load_net(self.scenario_dir)
            self.missions = [plan_mission(net, missions[a]) if a in missions else None for a in self.agent_ids]
            self.missions += [None] * num_social
            slots = self.N + num_social
            for i, m in enumerate(self.missions):
                if m is not None:
                    x, y, h = m.spawn_pose()
                    spawns.reshape(spawns.shape[0], num_envs, slots, 4)[:, :, i] = (x, y, h, 0.0)
        self.sim = BatchedSim(self.cm, self.cfg, device=device, spawns=spawns, seed=seed, social_spawns=where,
                              vias=self.vias, missions=self.missions)
        road_ids = [self.cm.road_ids[r] for r in self.cm.lane_road]
        vehicle_names = self.agent_ids + [f"social-{k}" for k in range(num_social)]
        self.builder = ObservationBuilder(
            self.cm.lane_ids, road_ids, vehicle_names, waypoints=self.cfg.waypoints, neighbors=self.cfg.neighbors,
            accelerometer=self.cfg.accelerometer, ogm=first.ogm or None, dagm=first.drivable_area_grid_map or None,
            lidar_rays=base_rays(first.lidar.sensor_params) if first.lidar else None, dt=dt, vias=self.vias,
            road_waypoints=bool(first.road_waypoints), missions=self.missions)
        self._was_reset = False
        self._destroyed = False

    @property
    def step_count(self) -> np.ndarray:
        """Ticks since each env's last reset, read from the device (``smx_state.env_ticks``): the one clock
        of the batch, also across ``reset_dense(env_mask)`` and the in-launch auto-reset."""
        return self.sim.env_ticks.cpu().numpy().astype(np.int64)

    # ------------------------------------------------------------------ dense (fast) path
    def reset_dense(self, env_mask=None):
        self._check_alive()
        out = self.sim.reset(env_mask)
        self._was_reset = True
        return out

    def step_dense(self, actions):
        """``actions``: int8 tensor [E, N] of lane-action codes (NO_ACTION = -1)."""
        self._check_alive()
        if not self._was_reset:
            raise SMARTSNotSetupError("Must call reset() or setup() before stepping.")
        return self.sim.step(actions)

    # ------------------------------------------------------------------ object path
    def step_actions(self, per_env_actions: Sequence[Dict[str, Any]]):
        """Per-env ``{agent_id: action}`` dicts -> one device tick (the reference's
        ``SMARTS.step(agent_actions)``); returns the dict of output tensors."""
        import torch

        from ..engine import pack_trajectory

        if self.interface.action is not ActionSpaceType.Trajectory:
            acts = self.encode_actions(per_env_actions)
            return self.step_dense(torch.from_numpy(acts).to(self.sim.device))
        # ActionSpaceType.Trajectory: (xs, ys, headings, speeds) per agent (controllers/__init__.py:104-110)
        slots = self.N + self.num_social
        packed = np.zeros((self.E, slots, 4, nat.TRAJ_COLS), dtype=np.float64)
        counts = np.zeros((self.E, slots), dtype=np.int32)
        for e, agent_actions in enumerate(per_env_actions):
            assert isinstance(agent_actions, dict) and all(isinstance(k, str) for k in agent_actions), \
                "Expected Dict[str, any]"  # hiway_env.py:232-234
            for agent_id, action in agent_actions.items():
                adapted = self.agent_specs[agent_id].action_adapter(action)
                i = self.agent_ids.index(agent_id)
                if adapted is None:
                    continue  # count 0 = no action
                packed[e, i], counts[e, i] = pack_trajectory(adapted)
        self._check_alive()
        if not self._was_reset:
            raise SMARTSNotSetupError("Must call reset() or setup() before stepping.")
        return self.sim.step_trajectory(torch.from_numpy(packed), torch.from_numpy(counts))

    def encode_actions(self, per_env_actions: Sequence[Dict[str, Any]]) -> np.ndarray:
        space = self.interface.action
        lane = space is ActionSpaceType.Lane or space is None
        slots = self.N + self.num_social
        if lane:
            acts = np.full((self.E, slots), NO_ACTION, dtype=np.int8)
        else:
            acts = np.full((self.E, slots, 3), np.nan, dtype=np.float32)  # NaN = no action
        for e, agent_actions in enumerate(per_env_actions):
            assert isinstance(agent_actions, dict) and all(isinstance(k, str) for k in agent_actions), \
                "Expected Dict[str, any]"  # hiway_env.py:232-234
            for agent_id, action in agent_actions.items():
                adapted = self.agent_specs[agent_id].action_adapter(action)
                i = self.agent_ids.index(agent_id)
                if adapted is None:
                    continue  # controllers/__init__.py:90-91: no action, no control call this tick
                if space is None:
                    raise ValueError("perform_action(action_space=None, ...) has failed: the interface has no action space")
                if lane:
                    acts[e, i] = encode_lane_action(adapted)
                else:
                    acts[e, i] = encode_float_action(space, adapted)
        return acts

    def host_rows(self, out) -> Dict[str, np.ndarray]:
        import torch

        torch.cuda.synchronize(self.sim.device)
        # the learner block is [2, E, N] (reward / done axis first) and exists for the device-side gather only
        rows = {k: v.cpu().numpy() for k, v in out.items() if k != "learner"}
        rows["env_ticks"] = self.sim.env_ticks.cpu().numpy()
        return rows

    PER_ENV_ROWS = ("env_done", "env_ticks")  # [E]; every other row is [E, N, ...]

    def observations(self, rows: Dict[str, np.ndarray], env: int, present: np.ndarray) -> Dict[str, Observation]:
        er = {k: v[env] for k, v in rows.items() if k not in self.PER_ENV_ROWS}
        t = int(rows["env_ticks"][env])  # device clock: already that of the new episode after an auto-reset
        elapsed = round(t * self.dt, 6)
        return {self.agent_ids[i]: self.builder.build(er, i, t, elapsed) for i in range(self.N) if present[i]}

    def final_observations(self, rows: Dict[str, np.ndarray], env: int, present: np.ndarray, step_count: int) -> Dict[str, Observation]:
        """The finishing tick's observation of the agents of an env that restarted inside the launch
        (``smx_outputs.final_*``: ego block, events, distance travelled — the low-dimensional part; the sensor rows of
        that tick are gone): what the reference hands back as ``info[agent]["env_obs"]`` (parallel_env.py:303-309)."""
        er = {k: v[env] for k, v in rows.items() if k not in self.PER_ENV_ROWS}
        for k in ("ego_pos", "ego_f32", "ego_lane", "events", "dist"):
            er[k] = rows["final_" + k][env]
        elapsed = round(step_count * self.dt, 6)
        return {self.agent_ids[i]: self.builder.build(er, i, step_count, elapsed, low_dimensional=True)
                for i in range(self.N) if present[i]}

    def close(self):
        if not self._destroyed:
            self.sim.close()
            self._destroyed = True

    def _check_alive(self):
        if self._destroyed:
            raise SMARTSDestroyedError("BUG: SMARTS was destroyed and is no longer usable")
