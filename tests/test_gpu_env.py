"""GPU: the HiWayEnv / ParallelEnv mirror end to end, modelled on the reference's own tests
(smarts/env/tests/test_hiway_env.py, test_parallel_env.py) plus parity of the object API with the
oracle on BASELINE.json configs[0] (scenarios/loop, 1 env x 4 Laner agents, seed 42)."""
import numpy as np
import pytest

import parity

pytestmark = pytest.mark.gpu

AGENT_ID = "Agent-007"
REWARD_EXPECTED = 3.14159
INFO_EXTRA_KEY = "__test_extra__"


def _adapted_spec(max_episode_steps=100):
    from smarts_amd.env import Agent, AgentInterface, AgentSpec, AgentType

    def observation_adapter(env_observation):  # test_hiway_env.py:42-53
        ego = env_observation.ego_vehicle_state
        wps = [path[0] for path in env_observation.waypoint_paths]
        closest_wp = min(wps, key=lambda wp: wp.dist_to(ego.position))
        return {"distance_from_center": closest_wp.signed_lateral_error(ego.position) / (closest_wp.lane_width * 0.5)}

    def reward_adapter(env_obs, env_reward):  # test_hiway_env.py:55-64
        assert -3 < env_reward < 3
        return REWARD_EXPECTED

    def info_adapter(env_obs, env_reward, env_info):
        env_info[INFO_EXTRA_KEY] = "blah"
        return env_info

    return AgentSpec(
        interface=AgentInterface.from_type(AgentType.Laner, max_episode_steps=max_episode_steps),
        agent_builder=lambda: Agent.from_function(lambda _: "KEEP_LANE"),
        observation_adapter=observation_adapter, reward_adapter=reward_adapter,
        action_adapter=lambda a: a.lower(), info_adapter=info_adapter,
    )


def test_hiway_env_adapters_and_episode_end():
    """test_hiway_env.py:93-131."""
    from smarts_amd.env import HiWayEnv, SMARTSNotSetupError

    spec = _adapted_spec(max_episode_steps=20)
    env = HiWayEnv(scenarios=["scenarios/loop"], agent_specs={AGENT_ID: spec}, headless=True, seed=42)
    with pytest.raises(SMARTSNotSetupError):
        env.step({AGENT_ID: "KEEP_LANE"})
    agent = spec.build_agent()
    for _ in range(2):
        observations = env.reset()
        assert set(observations) == {AGENT_ID} and "distance_from_center" in observations[AGENT_ID]
        dones = {"__all__": False}
        steps = 0
        while not dones["__all__"]:
            action = agent.act(observations[AGENT_ID])
            observations, rewards, dones, infos = env.step({AGENT_ID: action})
            assert rewards[AGENT_ID] == REWARD_EXPECTED
            assert infos[AGENT_ID][INFO_EXTRA_KEY] == "blah" and "score" in infos[AGENT_ID]
            assert infos[AGENT_ID]["env_obs"].ego_vehicle_state.id == f"{AGENT_ID}-vehicle"
            steps += 1
            assert steps <= 20
        assert dones[AGENT_ID] and infos[AGENT_ID]["env_obs"].events.reached_max_episode_steps
    assert env.scenario_log["scenario_map"] == "loop" and env.scenario_log["fixed_timestep_sec"] == 0.1
    env.close()


def test_hiway_env_matches_the_oracle_on_config0(nets, compiled_maps):
    """BASELINE.json configs[0]: loop, 1 env x 4 Laner agents."""
    from smarts_amd.engine import make_spawns
    from smarts_amd.env import Agent, AgentInterface, AgentSpec, AgentType, HiWayEnv
    from smarts_amd.env.core import sim_config_from_interface

    ids = [f"agent_{i}" for i in range(4)]
    itf = AgentInterface.from_type(AgentType.Laner, neighborhood_vehicles=True)
    specs = {a: AgentSpec(interface=itf, agent_builder=lambda: Agent.from_function(lambda _: "keep_lane")) for a in ids}
    env = HiWayEnv(scenarios=["scenarios/loop"], agent_specs=specs, seed=42)
    cm = compiled_maps("loop")
    cfg = sim_config_from_interface(itf, 1, 4, 0.1, False)
    # without missions every agent starts where the reference's Mission.random_endless_mission puts it under seed 42
    # (tests/golden/default_missions.npz, the reference's own draw; test_default_spawns_are_the_references below)
    from smarts_amd.missions import reference_spawn_table

    spawn = reference_spawn_table(nets("loop"), 1, 4, 42, episodes=4)
    assert not np.allclose(spawn[0], make_spawns(cm, 1, 4, episodes=4, seed=42)[0])
    ob = parity.OracleBatch(nets("loop"), cm, cfg, spawn[0])
    obs = env.reset()
    ref = ob.envs[0].reset_observe()
    script = ["keep_lane", "change_lane_left", "slow_down", "keep_lane", "change_lane_right", "keep_lane"]
    for t in range(12):
        for i, a in enumerate(ids):
            e, r = obs[a].ego_vehicle_state, ref[i]["ego"]
            assert np.allclose(e.position[:2], r["position"][:2], atol=1e-6), (t, a)
            assert e.lane_id == r["lane_id"] and e.lane_index == r["lane_index"]
            # a lone HiWayEnv keeps the reference's full paths (every path, lookahead + 1 waypoints)
            assert [p[0].lane_id for p in obs[a].waypoint_paths] == [p[0].lane_id for p in ref[i]["waypoint_paths"]]
            assert [len(p) for p in obs[a].waypoint_paths] == [len(p) for p in ref[i]["waypoint_paths"]]
            last, ref_last = obs[a].waypoint_paths[0][-1], ref[i]["waypoint_paths"][0][-1]
            assert np.allclose(last.pos, ref_last.pos[:2], atol=1e-6) and last.lane_id == ref_last.lane_id
            assert [v.id for v in obs[a].neighborhood_vehicle_states] == [f"agent_{nv['slot']}-vehicle" for nv in ref[i]["neighbors"]]
        acts = {a: script[(t + i) % len(script)] for i, a in enumerate(ids)}
        obs, rewards, dones, infos = env.step(acts)
        ref, ref_rew, ref_done = ob.envs[0].step([acts[a] for a in ids])
        assert {a: dones[a] for a in ids if a in dones} == {ids[i]: d for i, d in ref_done.items()}
        for i, a in enumerate(ids):
            if a in rewards:
                assert rewards[a] == pytest.approx(ref_rew[i], abs=1e-6)
        if dones["__all__"]:
            break
    env.close()


def _ctor(max_episode_steps=3):
    from smarts_amd.env import Agent, AgentInterface, AgentSpec, AgentType, HiWayEnv

    def make():
        spec = AgentSpec(interface=AgentInterface.from_type(AgentType.Laner, max_episode_steps=max_episode_steps),
                         agent_builder=lambda: Agent.from_function(lambda _: "keep_lane"))
        return HiWayEnv(scenarios=["scenarios/loop"], agent_specs={"Agent_0": spec, "Agent_1": spec}, seed=42)

    return make


def test_parallel_env_seed_reset_step():
    """test_parallel_env.py:81-160."""
    from smarts_amd.env import ParallelEnv

    with pytest.raises(TypeError):
        ParallelEnv(env_constructors=["not callable"], auto_reset=True)
    env = ParallelEnv(env_constructors=[_ctor()] * 2, auto_reset=True)
    assert env.batch_size == 2
    assert list(env.seed(7)) == [7, 8]
    batched = env.reset()
    assert len(batched) == 2 and all(set(o) == {"Agent_0", "Agent_1"} for o in batched)
    # different seeds -> different spawns
    assert not np.allclose(batched[0]["Agent_0"].ego_vehicle_state.position, batched[1]["Agent_0"].ego_vehicle_state.position)
    acts = {"Agent_0": "keep_lane", "Agent_1": "keep_lane"}
    obs, rewards, dones, infos = env.step([acts] * 2)
    for outputs, kind in ((rewards, float), (dones, bool)):
        assert len(outputs) == 2
        for o in outputs:
            o = dict(o)
            o.pop("__all__", None)
            assert set(o) == {"Agent_0", "Agent_1"} and all(isinstance(v, kind) for v in o.values())
    assert all(isinstance(i["Agent_0"]["score"], float) for i in infos)
    env.close()


@pytest.mark.parametrize("auto_reset", [True, False])
def test_parallel_env_sync_async_episodes(auto_reset):
    """test_parallel_env.py:166-189 with max_episode_steps = 3."""
    from smarts_amd.env import ParallelEnv

    env = ParallelEnv(env_constructors=[_ctor(3)] * 2, auto_reset=auto_reset)
    acts = [{"Agent_0": "keep_lane", "Agent_1": "keep_lane"}] * 2
    try:
        env.reset()
        _, _, dones, _ = env.step(acts)
        assert all(d["__all__"] is False for d in dones)
        obs, _, dones, _ = env.step(acts)
        assert all(d["__all__"] is True for d in dones)
        if auto_reset:
            # the observation handed back is the first of the next episode
            assert all(set(o) == {"Agent_0", "Agent_1"} for o in obs)
        _, _, dones, _ = env.step(acts if auto_reset else [{}] * 2)
        assert all(d["__all__"] is (not auto_reset) for d in dones)
        _, _, dones, _ = env.step(acts if auto_reset else [{}] * 2)
        assert all(d["__all__"] is True for d in dones)
    finally:
        env.close()


def test_parallel_env_dense_path_stays_on_device():
    import torch

    from smarts_amd.env import FormatObs, ParallelEnv

    env = ParallelEnv(env_constructors=[_ctor(50)] * 4, auto_reset=True)
    out = env.reset_dense()
    assert out["ego_pos"].is_cuda and out["ego_pos"].shape == (4, 2, 3) and out["wp_pos"].shape == (4, 2, 4, 20, 3)
    acts = torch.zeros((4, 2), dtype=torch.int8, device="cuda")
    p0 = out["ego_pos"].clone()
    out = env.step_dense(acts)
    torch.cuda.synchronize()
    # everybody moved (from a standstill: the reference's default missions start the agents at speed 0)
    assert (out["ego_pos"][..., :2] - p0[..., :2]).norm(dim=-1).min() > 1e-3
    rows = {k: v.cpu().numpy() for k, v in out.items()}
    s = FormatObs.from_rows(rows, 3, 1)
    assert s.waypoints["pos"].shape == (4, 20, 3) and s.ego["pos"].dtype == np.float64 and s.dist.dtype == np.float32
    env.close()


def test_hiway_env_with_scripted_social_traffic():
    from smarts_amd.env import Agent, AgentInterface, AgentSpec, AgentType, HiWayEnv

    spec = AgentSpec(interface=AgentInterface.from_type(AgentType.Laner, neighborhood_vehicles=True, max_episode_steps=30),
                     agent_builder=lambda: Agent.from_function(lambda _: "keep_lane"))
    env = HiWayEnv(scenarios=["scenarios/loop"], agent_specs={"A": spec, "B": spec}, seed=3, num_social=12)
    obs = env.reset()
    assert set(obs) == {"A", "B"}
    ids = {v.id for v in obs["A"].neighborhood_vehicle_states}
    assert "B-vehicle" in ids and any(i.startswith("social-") for i in ids) and len(ids) == 10  # the dense rows keep the first ten (format_obs.py:41)
    for _ in range(5):
        obs, rewards, dones, infos = env.step({"A": "keep_lane", "B": "slow_down"})
    assert set(rewards) == {"A", "B"} and not dones["__all__"]
    env.close()


def test_hiway_env_social_traffic_with_car_following():
    from smarts_amd.env import Agent, AgentInterface, AgentSpec, AgentType, HiWayEnv

    spec = AgentSpec(interface=AgentInterface.from_type(AgentType.Laner, neighborhood_vehicles=True, max_episode_steps=40),
                     agent_builder=lambda: Agent.from_function(lambda _: "slow_down"))
    env = HiWayEnv(scenarios=["scenarios/loop"], agent_specs={AGENT_ID: spec}, headless=True, seed=9, num_social=12,
                   social_model="idm")
    obs = env.reset()
    speeds = []
    for _ in range(30):
        obs, _, dones, _ = env.step({AGENT_ID: "slow_down"})
        speeds += [v.speed for v in obs[AGENT_ID].neighborhood_vehicle_states if v.id.startswith("social-")]
        if dones["__all__"]:
            break
    assert speeds and min(speeds) >= 0.0 and max(speeds) <= 16.67 + 1e-3
    env.close()


def test_hiway_env_tracker_agent_follows_its_waypoints():
    """AgentType.Tracker (ActionSpaceType.Trajectory): the agent sends back its first waypoint path
    with a speed profile, as the reference's examples do."""
    from smarts_amd.env import Agent, AgentInterface, AgentSpec, AgentType, HiWayEnv

    def act(obs):
        path = obs.waypoint_paths[0]
        return ([w.pos[0] for w in path], [w.pos[1] for w in path], [float(w.heading) for w in path], [9.0] * len(path))

    spec = AgentSpec(interface=AgentInterface.from_type(AgentType.Tracker, max_episode_steps=40),
                     agent_builder=lambda: Agent.from_function(act))
    env = HiWayEnv(scenarios=["scenarios/loop"], agent_specs={"T": spec}, seed=11)
    agent = spec.build_agent()
    obs = env.reset()
    dist = 0.0
    for _ in range(30):
        obs, rewards, dones, infos = env.step({"T": agent.act(obs["T"])})
        dist += rewards["T"]
        assert not obs["T"].events.off_road
    assert dist > 15.0 and 6.0 < obs["T"].ego_vehicle_state.speed < 12.0  # settles near the 9 m/s it asks for
    env.close()


def test_hiway_env_reports_mission_vias():
    """Observation.via_data (sensors.py:165-172): near vias sorted by distance, hits when passed at speed."""
    from smarts_amd.env import Agent, AgentInterface, AgentSpec, AgentType, HiWayEnv, Via

    spec = AgentSpec(interface=AgentInterface.from_type(AgentType.Laner, max_episode_steps=60),
                     agent_builder=lambda: Agent.from_function(lambda _: "keep_lane"))
    vias = [Via("edge-south-SN", 1, 30, 13, hit_distance=4.0), Via("edge-south-SN", 0, 45, 13, hit_distance=4.0),
            Via("edge-west-EW", 0, 20, 8)]
    env = HiWayEnv(scenarios=["scenarios/intersections/4lane"], agent_specs={"A": spec, "B": spec}, seed=5,
                   vias={"A": vias})
    obs = env.reset()
    assert obs["B"].via_data.near_via_points == [] and len(obs["A"].via_data.near_via_points) >= 1
    pos = obs["A"].ego_vehicle_state.position[:2]
    d = [np.hypot(p.position[0] - pos[0], p.position[1] - pos[1]) for p in obs["A"].via_data.near_via_points]
    assert d == sorted(d) and obs["A"].via_data.near_via_points[0].road_id.startswith("edge-")
    env.close()


def test_hiway_env_buddha_agent_sees_nothing_and_does_nothing():
    """AgentType.Buddha (agent_interface.py:308-309): no sensors, no action space; ``None`` actions
    reach no controller (controllers/__init__.py:90-91), so the vehicle just rolls on."""
    from smarts_amd.env import Agent, AgentInterface, AgentSpec, AgentType, HiWayEnv

    spec = AgentSpec(interface=AgentInterface.from_type(AgentType.Buddha, max_episode_steps=8),
                     agent_builder=lambda: Agent.from_function(lambda _: None))
    # (the benchmark's spawn table starts vehicles at the speed limit; the reference's default missions at a standstill)
    env = HiWayEnv(scenarios=["scenarios/loop"], agent_specs={AGENT_ID: spec}, headless=True, seed=3, spawns="synthetic")
    obs = env.reset()
    start = np.array(obs[AGENT_ID].ego_vehicle_state.position[:2])
    assert not obs[AGENT_ID].waypoint_paths and not obs[AGENT_ID].neighborhood_vehicle_states
    for _ in range(3):
        obs, rewards, dones, _ = env.step({AGENT_ID: None})
    moved = np.linalg.norm(np.array(obs[AGENT_ID].ego_vehicle_state.position[:2]) - start)
    assert moved > 1.0 and obs[AGENT_ID].ego_vehicle_state.steering == pytest.approx(0.0, abs=1e-6)
    with pytest.raises(ValueError):
        env.step({AGENT_ID: "keep_lane"})  # an action without an action space fails in the controller dispatch
    env.close()


def test_hiway_env_drivable_area_grid_map_and_std_obs():
    """AgentInterface.drivable_area_grid_map (agent_interface.py:29-38, 234-239) through the object API and
    FormatObs' ``dagm`` key (format_obs.py:393-398)."""
    from smarts_amd.env import Agent, AgentInterface, AgentSpec, AgentType, DrivableAreaGridMap, FormatObs, HiWayEnv

    itf = AgentInterface.from_type(AgentType.Laner, drivable_area_grid_map=DrivableAreaGridMap(64, 64, 50 / 64),
                                   neighborhood_vehicles=True)
    spec = AgentSpec(interface=itf, agent_builder=lambda: Agent.from_function(lambda _: "keep_lane"))
    env = HiWayEnv(scenarios=["scenarios/loop"], agent_specs={AGENT_ID: spec}, headless=True, seed=11)
    obs = env.reset()
    grid = obs[AGENT_ID].drivable_area_grid_map
    assert grid.data.shape == (64, 64, 1) and grid.data.dtype == np.uint8 and grid.metadata.resolution == 50 / 64
    assert grid.data[30:34, 30:34].max() == 255 and grid.metadata.camera_pos == tuple(obs[AGENT_ID].ego_vehicle_state.position)
    env.close()
    wrapped = FormatObs(HiWayEnv(scenarios=["scenarios/loop"], agent_specs={AGENT_ID: spec}, headless=True, seed=11))
    std = wrapped.reset()[AGENT_ID]
    assert std.dagm.shape == (64, 64, 1) and np.array_equal(std.dagm, grid.data)
    wrapped.close()


def test_parallel_env_object_api_with_five_envs_and_auto_reset():
    """The object API over more than two envs (round 1 sliced the [2, E, N] learner block by env and failed from
    the third env on), across an auto-reset: every env hands back its own agents' observations, the
    observation that follows an episode end is the first of the next episode and carries ITS clock
    (step_count / elapsed_sim_time restart with the env, read from the device's env_ticks)."""
    from smarts_amd.env import ParallelEnv

    env = ParallelEnv(env_constructors=[_ctor(3)] * 5, auto_reset=True, seed=11)
    acts = [{"Agent_0": "keep_lane", "Agent_1": "keep_lane"}] * 5
    try:
        first = env.reset()
        assert len(first) == 5
        pos0 = [tuple(o["Agent_0"].ego_vehicle_state.position) for o in first]
        assert len(set(pos0)) == 5  # five different worlds (seed + i)
        t0 = [o["Agent_0"].step_count for o in first]
        assert len(set(t0)) == 1
        obs, rew, dones, infos = env.step(acts)
        assert all(set(o) == {"Agent_0", "Agent_1"} for o in obs) and all(d["__all__"] is False for d in dones)
        assert [o["Agent_0"].step_count for o in obs] == [t0[0] + 1] * 5
        obs, rew, dones, infos = env.step(acts)  # max_episode_steps = 3: every agent ends here, every env restarts
        assert all(d["__all__"] is True for d in dones) and all(set(r) == {"Agent_0", "Agent_1"} for r in rew)
        assert [o["Agent_0"].step_count for o in obs] == t0  # first observation of the NEXT episode, its own clock
        assert [round(o["Agent_0"].elapsed_sim_time, 6) for o in obs] == [round(t0[0] * 0.1, 6)] * 5
        obs, rew, dones, infos = env.step(acts)
        assert all(d["__all__"] is False for d in dones)
    finally:
        env.close()


def test_lane_action_code_that_names_no_action_is_reported_at_sync(compiled_maps):
    """controllers/__init__.py:137-144 looks the Lane action up in a dict and raises on an unknown one.  The dense
    path takes codes: one outside -1..3 moves nothing (the agent is stepped as if it had sent no action) and the
    next smx_sync says so, once."""
    import torch

    from smarts_amd import _native as nat
    from smarts_amd.engine import BatchedSim, SimConfig

    for strategy in ("small", "large", "large_one_lane"):
        sim = BatchedSim(compiled_maps("loop"), SimConfig(num_envs=2, num_vehicles=2, launch_strategy=strategy))
        ref = BatchedSim(compiled_maps("loop"), SimConfig(num_envs=2, num_vehicles=2, launch_strategy=strategy))
        sim.reset(), ref.reset()
        bad = torch.tensor([[0, 7], [0, 0]], dtype=torch.int8, device="cuda")
        none = torch.tensor([[0, -1], [0, 0]], dtype=torch.int8, device="cuda")
        sim.step(bad), ref.step(none)
        with pytest.raises(nat.SmxError, match="Lane action code"):
            sim.sync()
        sim.sync()  # reported once
        ref.sync()
        assert torch.equal(sim.state, ref.state)
        sim.close(), ref.close()
        # a wrong shape is refused before anything is enqueued (and before the learner block flips)
        sim = BatchedSim(compiled_maps("loop"), SimConfig(num_envs=2, num_vehicles=2, launch_strategy=strategy))
        sim.reset()
        k = sim._learner_k
        with pytest.raises(ValueError):
            sim.step(torch.zeros((2, 3), dtype=torch.int8, device="cuda"))
        assert sim._learner_k == k
        sim.close()


def test_default_spawns_are_the_references(nets):
    """BASELINE configs[0] ("parity seed"): hiway-v0 on scenarios/loop without missions.pkl starts its four agents on
    Mission.random_endless_mission x 4 drawn after smarts.core.seed(42) and the scenario rolls (plan.py:225-249,
    sumo_road_network.py:803-810, scenario.py:211-214); the fixture holds the reference's own draw.  The first reset of
    the env must put the vehicles half a chassis length behind those starts (Pose.from_front_bumper), standing."""
    import math
    import os

    from smarts_amd.env import Agent, AgentInterface, AgentSpec, AgentType, HiWayEnv

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "default_missions.npz"))
    ids = [f"agent_{i}" for i in range(4)]
    itf = AgentInterface.from_type(AgentType.Laner)
    specs = {a: AgentSpec(interface=itf, agent_builder=lambda: Agent.from_function(lambda _: "keep_lane")) for a in ids}
    env = HiWayEnv(scenarios=["scenarios/loop"], agent_specs=specs, seed=42)
    obs = env.reset()
    for i, a in enumerate(ids):
        start, heading = g["loop_rolls3_position"][i], float(g["loop_rolls3_heading"][i])
        e = obs[a].ego_vehicle_state
        ang = (heading + math.pi * 0.5) % (2 * math.pi)
        centre = (start[0] - math.cos(ang) * 0.5 * 3.68, start[1] - math.sin(ang) * 0.5 * 3.68)
        assert np.allclose(e.position[:2], centre, atol=1e-9), (a, e.position, centre)
        assert abs(float(e.heading) - heading) < 1e-6 and abs(float(e.speed)) < 1e-9
    env.close()


def test_parallel_env_auto_reset_keeps_the_final_observation():
    """smarts/env/wrappers/parallel_env.py:303-309 + smarts/env/hiway_env.py:243-246: under auto_reset the
    observation returned with dones["__all__"] is the next episode's first one, and the finishing tick's observation
    travels in info[agent]["env_obs"] (its score in info["score"]).  The device keeps that tick's low-dimensional rows
    (smx_outputs.final_*): they must equal what the same envs report without auto_reset."""
    from smarts_amd.env import ParallelEnv

    acts = [{"Agent_0": "keep_lane", "Agent_1": "change_lane_left"}] * 3
    finals = {}
    for auto_reset in (True, False):
        env = ParallelEnv(env_constructors=[_ctor(3)] * 3, auto_reset=auto_reset, seed=7)
        try:
            first = env.reset()
            env.step(acts)
            obs, rewards, dones, infos = env.step(acts)
            assert all(d["__all__"] for d in dones)
            finals[auto_reset] = (obs, rewards, infos)
            if auto_reset:
                # the observations are those of the new episode (episode 1 of the spawn table): standing vehicles
                for e in range(3):
                    for a in ("Agent_0", "Agent_1"):
                        assert obs[e][a].ego_vehicle_state.speed == pytest.approx(0.0, abs=1e-9)
                        assert not np.allclose(obs[e][a].ego_vehicle_state.position, first[e][a].ego_vehicle_state.position)
        finally:
            env.close()
    (_, rew_a, info_a), (obs_b, rew_b, info_b) = finals[True], finals[False]
    for e in range(3):
        for a in ("Agent_0", "Agent_1"):
            last, ref = info_a[e][a]["env_obs"], obs_b[e][a]
            assert last is not None and last.events.reached_max_episode_steps
            assert np.array_equal(last.ego_vehicle_state.position, ref.ego_vehicle_state.position)
            assert float(last.ego_vehicle_state.heading) == float(ref.ego_vehicle_state.heading)
            assert last.ego_vehicle_state.speed == ref.ego_vehicle_state.speed
            assert last.ego_vehicle_state.lane_id == ref.ego_vehicle_state.lane_id
            assert last.events == ref.events
            assert info_a[e][a]["score"] == info_b[e][a]["score"] and rew_a[e][a] == rew_b[e][a]
