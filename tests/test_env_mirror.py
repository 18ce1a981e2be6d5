"""Host-side mirror of the reference's env surface (smarts_amd/env): interface presets, AgentSpec,
Observation construction from dense rows, lane_ttc / StdObs against outputs of the reference's own
functions (tests/golden/std_obs.npz, produced by tests/golden/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest

import parity
from conftest import GOLDEN
from smarts_amd.env import (OGM, ActionSpaceType, Agent, AgentInterface, AgentSpec, AgentType, DoneCriteria, Heading,
                            Lidar, NeighborhoodVehicles, Waypoints)
from smarts_amd.env import core as env_core
from smarts_amd.env.custom_observations import lane_ttc
from smarts_amd.env.format_obs import FormatObs, std_obs
from smarts_amd.env.observations import ObservationBuilder


def test_agent_interface_presets_follow_the_reference():
    # agent_interface.py:299-396
    laner = AgentInterface.from_type(AgentType.Laner, max_episode_steps=7)
    assert laner.action is ActionSpaceType.Lane and laner.waypoints == Waypoints(lookahead=32)
    assert laner.neighborhood_vehicles is False and laner.max_episode_steps == 7 and laner.accelerometer
    full = AgentInterface.from_type(AgentType.Full)
    assert full.ogm == OGM(256, 256, 50 / 256) and full.lidar == Lidar() and full.rgb and full.drivable_area_grid_map
    assert full.action is ActionSpaceType.Continuous
    std = AgentInterface.from_type(AgentType.Standard)
    assert std.action is ActionSpaceType.ActuatorDynamic and std.neighborhood_vehicles == NeighborhoodVehicles(None)
    assert AgentInterface.from_type(AgentType.Buddha).action is None
    assert AgentInterface(waypoints=True).replace(waypoints=Waypoints(8)).waypoints.lookahead == 8
    assert DoneCriteria() == DoneCriteria(collision=True, off_road=True, off_route=True, on_shoulder=False,
                                          wrong_way=False, not_moving=False, agents_alive=None)
    assert len(Lidar().sensor_params.laser_angles) == 50  # BasicLidar, lidar_sensor_params.py:48-56


def test_unsupported_interfaces_fail_loudly():
    with pytest.raises(NotImplementedError):
        AgentInterface.from_type(AgentType.Full).validate_for_device()  # rgb / dagm
    with pytest.raises(NotImplementedError):
        AgentInterface.from_type(AgentType.MPCTracker).validate_for_device()  # MPC action space
    for t in (AgentType.Standard, AgentType.StandardWithAbsoluteSteering, AgentType.LanerWithSpeed, AgentType.Loner,
              AgentType.Tracker):
        AgentInterface.from_type(t).validate_for_device()
    with pytest.raises(NotImplementedError):
        AgentInterface.from_type(AgentType.Laner, rgb=True).validate_for_device()
    AgentInterface.from_type(AgentType.Laner, ogm=True, lidar=True, neighborhood_vehicles=True).validate_for_device()


def test_agent_spec_builds_agents_like_the_reference():
    # agent_spec.py:84-118
    spec = AgentSpec(interface=AgentInterface.from_type(AgentType.Laner),
                     agent_builder=lambda: Agent.from_function(lambda _: "keep_lane"))
    assert spec.build_agent().act(None) == "keep_lane"

    class P(Agent):
        def __init__(self, a, b=2):
            self.a, self.b = a, b

    assert AgentSpec(agent_builder=P, agent_params=(1, 3)).build_agent().b == 3
    assert AgentSpec(agent_builder=P, agent_params={"a": 5, "zzz": 1}).build_agent().a == 5  # extra keys dropped
    assert AgentSpec(agent_builder=P, agent_params=9).build_agent().a == 9
    with pytest.raises(ValueError):
        AgentSpec().build_agent()
    assert spec.reward_adapter(None, 1.5) == 1.5 and spec.info_adapter(None, 0, {"k": 1}) == {"k": 1}


def test_lane_action_encoding():
    assert [env_core.encode_lane_action(a) for a in ("keep_lane", "slow_down", "change_lane_left", "change_lane_right")] == [0, 1, 2, 3]
    with pytest.raises(KeyError):
        env_core.encode_lane_action("fly")
    with pytest.raises(TypeError):
        env_core.encode_lane_action(3)


def test_float_action_encoding():
    A = ActionSpaceType
    assert env_core.encode_float_action(A.Continuous, (0.5, 0.0, -0.25)) == [0.5, 0.0, -0.25]
    assert env_core.encode_float_action(A.LaneWithContinuousSpeed, (12.0, -1)) == [12.0, -1.0, 0.0]
    with pytest.raises(ValueError):
        env_core.encode_float_action(A.ActuatorDynamic, (0.1, 0.2))


def test_trajectory_packing():
    from smarts_amd.engine import pack_trajectory

    xs = list(range(25))
    packed, n = pack_trajectory((xs, [2 * x for x in xs], [0.1] * 25, [5.0 + x for x in xs]))
    assert n == 25 and packed.shape == (4, 11)
    assert packed[0, :10].tolist() == xs[:10] and packed[0, 10] == 24 and packed[3, 10] == 29.0
    short, n = pack_trajectory(([1.0, 2.0], [0.0, 0.0], [0.0, 0.0], [3.0, 4.0]))
    assert n == 2 and short[0, :2].tolist() == [1.0, 2.0] and short[3, 10] == 4.0
    with pytest.raises(ValueError):
        pack_trajectory(([1.0], [1.0, 2.0], [0.0], [0.0]))


def test_scenario_resolution():
    d = env_core.resolve_scenario("scenarios/loop")
    assert os.path.exists(os.path.join(d, "map.smxnet.json.gz"))
    assert env_core.resolve_scenario("scenarios/intersections/4lane").endswith("4lane")
    with pytest.raises(FileNotFoundError):
        env_core.resolve_scenario("scenarios/does_not_exist")


def test_sim_config_from_interface():
    itf = AgentInterface.from_type(AgentType.Laner, neighborhood_vehicles=NeighborhoodVehicles(radius=30.0),
                                   ogm=OGM(64, 64, 50 / 64), max_episode_steps=11,
                                   done_criteria=DoneCriteria(on_shoulder=True, collision=False))
    cfg = env_core.sim_config_from_interface(itf, 3, 5, 0.1, True)
    assert (cfg.num_envs, cfg.num_vehicles, cfg.wp_lookahead, cfg.wp_paths, cfg.wp_len) == (3, 5, 32, 4, 20)
    assert cfg.neighbors and cfg.nb_radius == 30.0 and cfg.ogm and cfg.ogm_width == 64 and cfg.lidar is None
    assert cfg.done_on_shoulder and not cfg.done_collision and cfg.max_episode_steps == 11 and cfg.auto_reset
    from smarts_amd.env.agent_interface import AgentsAliveDoneCriteria, AgentsListAlive

    alive = AgentInterface.from_type(AgentType.Laner, done_criteria=DoneCriteria(agents_alive=AgentsAliveDoneCriteria(
        minimum_ego_agents_alive=2, agent_lists_alive=[AgentsListAlive(agents_list=["b", "zz", "c"], minimum_agents_alive_in_list=1)])))
    ca = env_core.sim_config_from_interface(alive, 1, 3, 0.1, False, agent_ids=["a", "b", "c"])
    assert ca.alive_min_ego == 2 and ca.alive_min_total is None and ca.alive_lists == (([1, 2], 1),)
    short = env_core.sim_config_from_interface(AgentInterface.from_type(AgentType.Laner, waypoints=Waypoints(8)), 1, 1, 0.1, False)
    assert short.wp_len == 9
    # None = what the reference's Observation holds: whole paths
    full = env_core.sim_config_from_interface(AgentInterface.from_type(AgentType.Laner), 1, 1, 0.1, False, waypoint_window=None)
    assert (full.wp_paths, full.wp_len) == (env_core.FULL_WINDOW_PATHS, 33)


def test_heading_wraps_like_the_reference():
    # coordinates.py:175-184
    assert Heading(3 * np.pi / 2) == pytest.approx(-np.pi / 2)
    assert Heading(np.pi) == pytest.approx(np.pi)
    assert Heading(-np.pi) == pytest.approx(np.pi)
    assert Heading(0.3).relative_to(Heading(-0.2)) == pytest.approx(0.5)


@pytest.fixture(scope="module")
def loop_rollout(nets, compiled_maps):
    from smarts_amd.engine import SimConfig, make_spawns

    cm = compiled_maps("loop")
    E, N = 1, 6
    cfg = SimConfig(num_envs=E, num_vehicles=N, neighbors=True, nb_radius=50.0)
    spawns = make_spawns(cm, E, N, episodes=1, seed=5)
    ob = parity.OracleBatch(nets("loop"), cm, cfg, spawns[0])
    ob.reset_observe()
    raw, rew, dones = ob.envs[0].step(["keep_lane"] * N)
    rows = parity.pack(cfg, ob.lane_no, N, raw, rew, dones)
    return cm, cfg, N, raw, rows


def test_observation_builder_matches_the_oracle_objects(loop_rollout):
    cm, cfg, N, raw, rows = loop_rollout
    ids = [f"a{i}" for i in range(N)]
    b = ObservationBuilder(cm.lane_ids, [cm.road_ids[r] for r in cm.lane_road], ids, waypoints=True, neighbors=True,
                           accelerometer=True, dt=0.1)
    for i in range(N):
        o, ref = b.build(rows, i, 3, 0.3), raw[i]
        e = o.ego_vehicle_state
        assert e.id == f"a{i}-vehicle" and e.lane_id == ref["ego"]["lane_id"] and e.road_id == ref["ego"]["road_id"]
        assert e.lane_index == ref["ego"]["lane_index"]
        assert np.allclose(e.position, ref["ego"]["position"]) and e.speed == pytest.approx(ref["ego"]["speed"], rel=1e-6)
        assert e.bounding_box.as_lwh == pytest.approx((3.68, 1.47, 1.0))
        assert o.distance_travelled == pytest.approx(ref["distance_travelled"])
        assert len(o.waypoint_paths) == min(len(ref["waypoint_paths"]), 4)
        for p, rp in zip(o.waypoint_paths, ref["waypoint_paths"]):
            assert len(p) == 20  # the StdObs window of a 33-waypoint path
            assert [w.lane_id for w in p] == [w.lane_id for w in rp[:20]]
            assert np.allclose([w.pos for w in p], [w.pos for w in rp[:20]])
        assert [v.id for v in o.neighborhood_vehicle_states] == [f"a{nv['slot']}-vehicle" for nv in ref["neighbors"]]
        assert [v.lane_id for v in o.neighborhood_vehicle_states] == [nv["lane_id"] for nv in ref["neighbors"]]
        assert o.events.off_road == ref["events"]["off_road"] and not o.events.collisions
        assert o.via_data.near_via_points == [] and o.occupancy_grid_map is None and o.lidar_point_cloud is None


def test_lane_ttc_and_std_obs_match_the_reference_functions(compiled_maps):
    """custom_observations.py:148-280 and format_obs.py:401-603, run by gen_golden.py on the same
    dense rows."""
    g = np.load(os.path.join(GOLDEN, "std_obs.npz"))
    cm = compiled_maps("loop")
    N = 8
    b = ObservationBuilder(cm.lane_ids, [cm.road_ids[r] for r in cm.lane_road], [f"agent_{i}" for i in range(N)],
                           waypoints=True, neighbors=True, accelerometer=True, dt=0.1)
    checked = 0
    for t in range(int(g["n_ticks"])):
        rows = {k[len(f"t{t}_in_"):]: g[k] for k in g.files if k.startswith(f"t{t}_in_")}
        for i in range(N):
            if f"t{t}_a{i}_lanettc_ego_ttc" not in g.files:
                continue
            o = b.build(rows, i, 0, 0.0)
            ttc = lane_ttc(o)
            for k, v in ttc.items():
                assert np.array_equal(np.asarray(v, dtype=np.float64), g[f"t{t}_a{i}_lanettc_{k}"]), (t, i, k)
            s = std_obs(o)
            for k, v in s.waypoints.items():
                assert np.array_equal(v, g[f"t{t}_a{i}_wp_{k}"]) and v.dtype == g[f"t{t}_a{i}_wp_{k}"].dtype, (t, i, k)
            if s.neighbors is not None:
                for k, v in s.neighbors.items():
                    assert np.array_equal(v, g[f"t{t}_a{i}_nb_{k}"]) and v.dtype == g[f"t{t}_a{i}_nb_{k}"].dtype
            else:
                assert f"t{t}_a{i}_nb_pos" not in g.files
            if s.ttc is not None:
                for k, v in s.ttc.items():
                    assert np.array_equal(np.asarray(v), g[f"t{t}_a{i}_ttc_{k}"]), (t, i, k)
            for k, v in s.ego.items():
                assert np.array_equal(np.asarray(v), g[f"t{t}_a{i}_ego_{k}"]), (t, i, k)
            # the object-free slicing of the same rows agrees with the object route
            d = FormatObs.from_rows({k: v[None] for k, v in rows.items()}, 0, i)
            for k in ("heading", "lane_index", "lane_width", "pos", "speed_limit"):
                assert np.array_equal(d.waypoints[k], s.waypoints[k])
            assert np.array_equal(d.ego["pos"], s.ego["pos"]) and d.ego["speed"] == s.ego["speed"]
            checked += 1
    assert checked >= 16


def test_make_mirrors_the_registered_id():
    # smarts/env/__init__.py:22-25 registers "hiway-v0"; gym.make("smarts.env:hiway-v0", ...)
    from smarts_amd import env as env_pkg

    spec = AgentSpec(interface=AgentInterface.from_type(AgentType.Laner))
    e = env_pkg.make("smarts.env:hiway-v0", scenarios=["scenarios/loop"], agent_specs={"A": spec}, headless=True, seed=7)
    assert isinstance(e, env_pkg.HiWayEnv) and e.agent_specs == {"A": spec} and e.seed(7) == 7
    e.close()
    with pytest.raises(ValueError):
        env_pkg.make("smarts.env:highway-v9", scenarios=["scenarios/loop"], agent_specs={"A": spec})


def test_buddha_interface_is_accepted_without_an_action_space():
    itf = AgentInterface.from_type(AgentType.Buddha)
    itf.validate_for_device()
    cfg = env_core.sim_config_from_interface(itf, 2, 3, 0.1, False)
    assert cfg.action_space == "Lane" and not cfg.waypoints and not cfg.neighbors
