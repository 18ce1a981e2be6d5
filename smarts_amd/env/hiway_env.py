"""``HiWayEnv`` mirror (reference ``smarts/env/hiway_env.py:36-291``): the same constructor
arguments, ``step`` / ``reset`` / ``seed`` / ``close`` contracts and dict-of-agent-id returns, with
the simulation running on the MI355X path.  Arguments that configure subsystems outside the hot
path (Envision, Visdom, SUMO traffic, zoo) are accepted for call compatibility and must be left at
values that disable them.
"""
from __future__ import annotations

import logging
import warnings
from typing import Any, Dict, Optional, Sequence, Tuple, Union

import numpy as np

from .agent import AgentSpec
from .core import BatchCore, SMARTSNotSetupError, resolve_scenario
from .observations import Observation


class HiWayEnv:
    """One environment instance: ``len(agent_specs)`` ego agents on one scenario map."""

    metadata = {"render.modes": ["human"]}

    def __init__(
        self,
        scenarios: Sequence[str],
        agent_specs: Dict[str, AgentSpec],
        sim_name: Optional[str] = None,
        shuffle_scenarios: bool = True,
        headless: bool = True,
        visdom: bool = False,
        fixed_timestep_sec: Optional[float] = None,
        seed: int = 42,
        num_external_sumo_clients: int = 0,
        sumo_headless: bool = True,
        sumo_port: Optional[str] = None,
        sumo_auto_start: bool = True,
        endless_traffic: bool = True,
        envision_endpoint: Optional[str] = None,
        envision_record_data_replay_path: Optional[str] = None,
        zoo_addrs: Optional[str] = None,
        timestep_sec: Optional[float] = None,  # deprecated alias (hiway_env.py:107-113)
        device: str = "cuda:0",
        waypoint_window: Optional[Tuple[int, int]] = None,
        num_social: int = 0,
        social_model: str = "constant",
        vias: Optional[Dict[str, Sequence]] = None,
        missions: Optional[Union[Dict[str, Any], str]] = None,
        spawns: str = "reference",
    ):
        self._log = logging.getLogger(self.__class__.__name__)
        if not headless or envision_record_data_replay_path or envision_endpoint:
            raise NotImplementedError("Envision visualisation is outside the accelerated path (use headless=True)")
        if visdom:
            raise NotImplementedError("Visdom is outside the accelerated path")
        if zoo_addrs or num_external_sumo_clients:
            raise NotImplementedError("zoo agents / external SUMO clients are outside the accelerated path")
        if timestep_sec and not fixed_timestep_sec:
            warnings.warn("timestep_sec has been deprecated in favor of fixed_timestep_sec.  Please update your code.",
                          category=DeprecationWarning)
        if not fixed_timestep_sec:
            fixed_timestep_sec = timestep_sec or 0.1
        if len(scenarios) != 1:
            raise NotImplementedError("one scenario per accelerated env (the map tables are loaded once)")
        self._scenario = resolve_scenario(scenarios[0])
        self._agent_specs = agent_specs
        self._dt = float(fixed_timestep_sec)
        self._device = device
        # (paths, waypoints per path) kept per agent; None = whole paths as in the reference's Observation
        # (ParallelEnv falls back to the StdObs window (4, 20) for its dense rows)
        self._waypoint_window = waypoint_window
        # scripted social traffic (the accelerated path's stand-in for the scenario's SUMO flows)
        self._num_social = int(num_social)
        self._social_model = social_model  # "constant" | "idm" (include/smx.h SMX_SOCIAL_*)
        # mission via points per agent id (smarts_amd.vias.Via = sstudio's Via); the reference reads them
        # from the scenario's missions, which this path does not parse
        self._vias = dict(vias) if vias else None
        # fixed-route missions per agent id (smarts_amd.missions.Mission = sstudio's Mission / Route, or the path
        # of a missions JSON): the agent starts at the route's begin, is held to the route (waypoints, off_route,
        # trip meter) and ends at its goal (reached_goal); agents without one drive endless missions
        if isinstance(missions, str):
            from ..missions import load_missions

            missions = load_missions(missions)
        self._missions = dict(missions) if missions else None
        # where agents without a mission start: "reference" = a random endless mission each, as hiway-v0 draws them
        # (missions.reference_spawn_table); "synthetic" = the benchmark's spawn table (engine.make_spawns)
        self._spawns = spawns
        self._shuffle_scenarios = bool(shuffle_scenarios)
        self._dones_registered = 0
        self._core: Optional[BatchCore] = None
        self._seed = seed
        self._closed = False
        for spec in agent_specs.values():
            if spec.interface is None:
                raise ValueError("every AgentSpec needs an interface")
            spec.interface.validate_for_device()

    # ------------------------------------------------------------------ properties
    @property
    def agent_specs(self) -> Dict[str, AgentSpec]:
        return self._agent_specs

    @property
    def scenario_log(self) -> Dict[str, Union[float, str]]:
        """hiway_env.py:184-202."""
        import os

        return {
            "fixed_timestep_sec": self._dt,
            "scenario_map": os.path.basename(self._scenario),
            "scenario_routes": "",
            "mission_hash": str(hash(frozenset(self._agent_specs.keys()))),
        }

    def signature(self):
        """What must agree for envs to share one device batch (ParallelEnv)."""
        specs = self._agent_specs
        return (self._scenario, tuple(specs.keys()), tuple(repr(s.interface) for s in specs.values()), self._dt,
                self._waypoint_window, self._num_social, self._social_model, repr(self._vias), repr(self._missions),
                self._spawns, self._shuffle_scenarios)

    def seed(self, seed: int) -> int:
        """hiway_env.py:204-214.  Takes effect at the next ``reset`` that (re)builds the spawn table."""
        if seed != self._seed and self._core is not None:
            self._core.close()
            self._core = None
        self._seed = seed
        return seed

    # ------------------------------------------------------------------ gym surface
    def _ensure_core(self) -> BatchCore:
        if self._closed:
            from .core import SMARTSDestroyedError

            raise SMARTSDestroyedError("BUG: SMARTS was destroyed and is no longer usable")
        if self._core is None:
            self._core = BatchCore(self._scenario, self._agent_specs, num_envs=1, dt=self._dt, seed=self._seed,
                                   auto_reset=False, device=self._device, waypoint_window=self._waypoint_window,
                                   num_social=self._num_social, vias=self._vias, social_model=self._social_model,
                                   missions=self._missions, spawns=self._spawns, shuffle_scenarios=self._shuffle_scenarios)
        return self._core

    def step(self, agent_actions) -> Tuple[Dict[str, Observation], Dict[str, float], Dict[str, bool], Dict[str, Any]]:
        """hiway_env.py:216-263."""
        if self._core is None or not self._core._was_reset:
            if self._closed:
                self._ensure_core()
            raise SMARTSNotSetupError("Must call reset() or setup() before stepping.")
        core = self._core
        import torch

        rows = core.host_rows(core.step_actions([agent_actions]))
        observations, rewards, dones, infos = unpack_env(core, rows, 0)
        for done in dones.values():
            self._dones_registered += 1 if done else 0
        dones["__all__"] = self._dones_registered >= len(self._agent_specs)
        return observations, rewards, dones, infos

    def reset(self) -> Dict[str, Observation]:
        """hiway_env.py:265-281."""
        core = self._ensure_core()
        self._dones_registered = 0
        rows = core.host_rows(core.reset_dense())
        present = rows["active"][0].astype(bool)
        env_obs = core.observations(rows, 0, present)
        return {aid: self._agent_specs[aid].observation_adapter(obs) for aid, obs in env_obs.items()}

    def render(self, mode="human"):
        """Does nothing (hiway_env.py:283-285)."""

    def close(self):
        """hiway_env.py:287-291."""
        if self._core is not None:
            self._core.close()
            self._core = None
        self._closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def unpack_env(core: BatchCore, rows: Dict[str, np.ndarray], env: int):
    """Dense rows of one env -> the reference's four dicts (hiway_env.py:239-256): agents that acted
    this tick are present, including the ones that finished on it."""
    done_row = rows["done"][env].astype(bool)
    present = rows["active"][env].astype(bool) | done_row
    raw_obs = core.observations(rows, env, present)
    observations, rewards, dones, infos = {}, {}, {}, {}
    for i, aid in enumerate(core.agent_ids):
        if not present[i]:
            continue
        spec = core.agent_specs[aid]
        obs = raw_obs[aid]
        reward = float(rows["reward"][env, i])
        info = {"score": float(rows["dist"][env, i]), "env_obs": obs}  # agent_manager.py:233-234
        rewards[aid] = spec.reward_adapter(obs, reward)
        observations[aid] = spec.observation_adapter(obs)
        infos[aid] = spec.info_adapter(obs, reward, info)
        dones[aid] = bool(done_row[i])
    return observations, rewards, dones, infos
