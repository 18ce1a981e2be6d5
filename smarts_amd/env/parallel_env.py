"""``ParallelEnv`` mirror (reference ``smarts/env/wrappers/parallel_env.py:49-264``).

The reference runs one process per environment and talks over pipes; here the environments the
constructors describe become ONE device batch (they must agree on scenario, agents, interface and
timestep), stepped by one launch sequence.  Same surface: ``batch_size``, ``seed`` -> ``[seed + i]``,
``reset`` -> sequence of per-env observation dicts, ``step(actions)`` -> four sequences of per-env
dicts, ``auto_reset``.  ``step_dense`` / ``reset_dense`` expose the tensors without any host copy.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Sequence, Tuple

import numpy as np

from .core import BatchCore
from .hiway_env import HiWayEnv, unpack_env

# ParallelEnv's product is the dense StdObs rows, so envs built without an explicit window keep the
# StdObs one (format_obs.py:42) here; a lone HiWayEnv keeps full paths for its Observation objects
STD_WAYPOINT_WINDOW = (4, 20)

EnvConstructor = Callable[[], HiWayEnv]


class ParallelEnv:
    def __init__(self, env_constructors: Sequence[EnvConstructor], auto_reset: bool, seed: int = 42,
                 device: str = "cuda:0"):
        if any(not callable(ctor) for ctor in env_constructors):
            raise TypeError(
                f"Found non-callable `env_constructors`. Expected `env_constructors` of type "
                f"`Sequence[Callable[[], gym.Env]]`, but got {env_constructors}).")
        envs = [ctor() for ctor in env_constructors]
        if not envs:
            raise ValueError("need at least one environment constructor")
        if any(not isinstance(e, HiWayEnv) for e in envs):
            raise TypeError("env constructors must build smarts_amd.env.HiWayEnv instances")
        sig = envs[0].signature()
        if any(e.signature() != sig for e in envs[1:]):
            # the reference raises ValueError when the spaces differ (parallel_env.py:160-188)
            raise ValueError("environments of one accelerated batch must share scenario, agents, interface and timestep")
        self._proto = envs[0]
        self._num_envs = len(envs)
        self._auto_reset = auto_reset
        self._device = device
        self._closed = False
        self._core: BatchCore = None
        self._dones_registered = np.zeros(self._num_envs, dtype=np.int64)
        self.seed(seed)

    @property
    def batch_size(self) -> int:
        return self._num_envs

    @property
    def agent_specs(self):
        return self._proto.agent_specs

    def seed(self, seed: int) -> Sequence[int]:
        """parallel_env.py:190-202: env i gets ``seed + i`` (its random stream: the missions of its agents,
        missions.reference_spawn_table; with ``spawns="synthetic"`` the spawn generator ``PCG64(seed + i)``)."""
        if self._core is not None:
            self._core.close()
        p = self._proto
        self._core = BatchCore(p._scenario, p.agent_specs, num_envs=self._num_envs, dt=p._dt, seed=seed,
                               auto_reset=self._auto_reset, device=self._device, waypoint_window=p._waypoint_window or STD_WAYPOINT_WINDOW,
                               num_social=p._num_social, vias=p._vias, social_model=p._social_model, missions=p._missions,
                               spawns=p._spawns, shuffle_scenarios=p._shuffle_scenarios)
        self._seed = seed
        return [seed + i for i in range(self._num_envs)]

    # ------------------------------------------------------------------ object API
    def reset(self) -> Sequence[Dict[str, Any]]:
        core = self._core
        rows = core.host_rows(core.reset_dense())
        self._dones_registered[:] = 0
        out = []
        for e in range(self._num_envs):
            obs = core.observations(rows, e, rows["active"][e].astype(bool))
            out.append({aid: core.agent_specs[aid].observation_adapter(o) for aid, o in obs.items()})
        return out

    def step(self, actions: Sequence[Dict[str, Any]]) -> Tuple[Sequence, Sequence, Sequence, Sequence]:
        """parallel_env.py:214-233 + the worker's auto-reset (:303-309): when an env reports
        ``dones["__all__"]`` and ``auto_reset`` is on, the observation returned for it is the first
        one of its next episode; rewards / dones / infos are those of the finishing tick."""
        import torch

        core = self._core
        if len(actions) != self._num_envs:
            raise ValueError(f"expected {self._num_envs} action dicts, got {len(actions)}")
        # the finishing tick's rows are overwritten by the in-launch auto-reset for envs that end; reward / done
        # survive (keep_reward_done) and so do the low-dimensional rows of the final observation (smx_outputs.final_*)
        ticks_before = core.step_count
        rows = core.host_rows(core.step_actions(actions))
        obs_b, rew_b, done_b, info_b = [], [], [], []
        for e in range(self._num_envs):
            env_done = bool(rows["env_done"][e])
            if env_done and self._auto_reset:
                done_row = rows["done"][e].astype(bool)
                first_obs = core.observations(rows, e, rows["active"][e].astype(bool))
                observations = {aid: core.agent_specs[aid].observation_adapter(o) for aid, o in first_obs.items()}
                rewards = {aid: float(rows["reward"][e, i]) for i, aid in enumerate(core.agent_ids) if done_row[i]}
                dones = {aid: True for i, aid in enumerate(core.agent_ids) if done_row[i]}
                # info["env_obs"]: the finishing tick's observation (parallel_env.py:303-309, hiway_env.py:243-246), its
                # ego block and events; info["score"]: the distance travelled as of that tick (agent_manager.py:233-234)
                last = core.final_observations(rows, e, done_row, int(ticks_before[e]) + 1)
                infos = {}
                for i, aid in enumerate(core.agent_ids):
                    if done_row[i]:
                        spec = core.agent_specs[aid]
                        info = {"score": float(rows["final_dist"][e, i]), "env_obs": last[aid]}
                        infos[aid] = spec.info_adapter(last[aid], float(rows["reward"][e, i]), info)
                        rewards[aid] = spec.reward_adapter(last[aid], rewards[aid])
                self._dones_registered[e] = 0
                dones["__all__"] = True
            else:
                observations, rewards, dones, infos = unpack_env(core, rows, e)
                self._dones_registered[e] += sum(1 for d in dones.values() if d)
                dones["__all__"] = bool(self._dones_registered[e] >= core.N)
            obs_b.append(observations)
            rew_b.append(rewards)
            done_b.append(dones)
            info_b.append(infos)
        return tuple(obs_b), tuple(rew_b), tuple(done_b), tuple(info_b)

    # ------------------------------------------------------------------ dense API (no host copy)
    def reset_dense(self):
        """Dict of device tensors in the StdObs layout, [E, N, ...] (include/smx.h smx_outputs)."""
        self._dones_registered[:] = 0
        return self._core.reset_dense()

    def step_dense(self, actions):
        """``actions``: int8 device tensor [E, N] (0 keep_lane, 1 slow_down, 2 change_lane_left,
        3 change_lane_right, -1 none).  Returns the same dict of device tensors, updated in place."""
        return self._core.step_dense(actions)

    def close(self, terminate: bool = False):
        if self._core is not None:
            self._core.close()
        self._closed = True

    def __del__(self):
        try:
            if not self._closed:
                self.close(terminate=True)
        except Exception:
            pass
