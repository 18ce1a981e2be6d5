"""GPU parity of fixed-route missions (smx_set_missions): routed waypoint paths, off_route / wrong_way /
reached_goal events and the trip meter's on-route rule.

* against the reference-generated fixture ``tests/golden/missions_<map>.npz`` directly (the reference's
  ``waypoint_paths(pose, 32, route)`` and ``Sensors._vehicle_is_off_route_and_wrong_way``): integers and flags
  exact, float64 <= 1e-9, float32 rows to float32 rounding;
* against the oracle on teacher-forced rollouts of missions planned by ``smarts_amd.missions`` (both launch
  forms; with and without the waypoints sensor): ``parity.compare`` tolerances (float64 1e-9).
"""
import os

import numpy as np
import pytest

import parity
import tie_sensitive
from conftest import GOLDEN
from test_gpu_golden import _host, _sim_at_poses, differing_waypoint_rows

pytestmark = pytest.mark.gpu

MAP_NAMES = ["loop", "4lane", "minicity"]


def _golden_routes(g):
    off = g["route_off"]
    return [[str(r) for r in g["route_roads"][off[k]:off[k + 1]]] for k in range(int(g["n_routes"]))]


@pytest.mark.parametrize("strategy", ["small", "large", "large_one_lane"])
@pytest.mark.parametrize("name", MAP_NAMES)
def test_routed_rows_and_route_events_equal_the_reference(name, strategy, compiled_maps):
    import torch

    from smarts_amd import _native as nat
    from smarts_amd.missions import PlannedMission

    cm = compiled_maps(name)
    g = np.load(os.path.join(GOLDEN, f"missions_{name}.npz"))
    lane_no = np.array([cm.lane_ids.index(str(l)) for l in g["lane_ids"]])
    routes = _golden_routes(g)
    P, W = 8, 33
    differing, off_seen = [], 0
    for k, roads in enumerate(routes):
        rows = np.flatnonzero(g["pose_route"] == k)
        assert len(rows) > 20
        # the fixture's own slice for these poses: path offsets rebased
        p_lo, p_hi = g["wp32_path_off"][rows[0]], g["wp32_path_off"][rows[-1] + 1]
        assert np.array_equal(rows, np.arange(rows[0], rows[-1] + 1))
        w = dict(path_off=g["wp32_path_off"][rows[0]:rows[-1] + 2] - p_lo,
                 wp_off=g["wp32_wp_off"][p_lo:p_hi + 1], x=g["wp32_x"], y=g["wp32_y"], heading=g["wp32_heading"],
                 lane=g["wp32_lane"], lane_index=g["wp32_lane_index"], width=g["wp32_width"], speed=g["wp32_speed"])
        sim = _sim_at_poses(cm, g["poses"][rows], wp_paths=P, wp_len=W, wp_lookahead=32, launch_strategy=strategy)
        # goal far from every pose: reached_goal stays off
        sim.set_missions([PlannedMission((0.0, 0.0), 0.0, (1e7, 1e7, 1.0), tuple(roads))])
        out = sim.reset()
        if strategy.startswith("large"):
            out = sim.step(torch.full((len(rows), 1), -1, dtype=torch.int8, device="cuda"))
        ev = _host(out["events"])[:, 0]
        differing += [int(rows[i]) for i in differing_waypoint_rows(out, w, lane_no, P, W)]
        sim.close()
        assert np.array_equal(ev[:, nat.EV_OFF_ROUTE], g["off_route"][rows]), (k, np.flatnonzero(ev[:, nat.EV_OFF_ROUTE] != g["off_route"][rows]))
        assert np.array_equal(ev[:, nat.EV_WRONG_WAY], g["wrong_way"][rows]), k
        assert not ev[:, nat.EV_REACHED_GOAL].any()
        off_seen += int(ev[:, nat.EV_OFF_ROUTE].sum())
    assert differing == tie_sensitive.ROUTE_WAYPOINTS[(name, 32)], differing
    assert off_seen > 0


def _missions_4lane(nets):
    from smarts_amd.missions import Mission, Route, plan_mission

    net = nets("4lane")
    specs = [
        Route(begin=("edge-north-NS", 0, 40), end=("edge-east-WE", 0, 30)),    # left turn through the junction
        Route(begin=("edge-west-WE", 1, 60), end=("edge-east-WE", 1, 12)),     # straight on; the goal lies close behind it
        Route(begin=("edge-south-SN", 0, 55), end=("edge-west-EW", 0, "max")),  # left turn
        Route(begin=("edge-east-EW", 1, 10), end=("edge-east-EW", 1, 40)),     # single-road route, goal 30 m ahead
    ]
    return [plan_mission(net, Mission(r)) for r in specs]


@pytest.mark.parametrize("strategy", ["small", "large", "large_one_lane"])
@pytest.mark.parametrize("waypoints", [True, False])
def test_fixed_route_rollout_against_the_oracle(strategy, waypoints, nets, compiled_maps):
    """Four agents drive their planned missions on the 4lane junction (one of them steered off its route by lane
    changes): waypoint rows along the route, the controller's routed paths (through the poses), off_route,
    reached_goal, the trip meter — every output, every tick, teacher-forced."""
    import torch

    from smarts_amd import _native as nat
    from smarts_amd.engine import BatchedSim, SimConfig

    cm = compiled_maps("4lane")
    missions = _missions_4lane(nets)
    E, N = 2, len(missions)
    spawns = np.zeros((1, E * N, 4))
    for e in range(E):
        for i, m in enumerate(missions):
            x, y, h = m.spawn_pose()
            spawns[0, e * N + i] = (x, y, h, 8.0 + e)
    cfg = SimConfig(num_envs=E, num_vehicles=N, neighbors=True, nb_radius=50.0, waypoints=waypoints,
                    launch_strategy=strategy)
    sim = BatchedSim(cm, cfg, spawns=spawns, missions=missions)
    ob = parity.OracleBatch(nets("4lane"), cm, cfg, spawns[0], missions=missions)

    def host(out):
        torch.cuda.synchronize()
        return {k: v.cpu().numpy().reshape((-1,) + tuple(v.shape[2:])) for k, v in out.items() if k != "env_done"}

    d, o = host(sim.reset()), ob.reset_observe()
    assert parity.compare(d, o, tol64=1e-9, tol32=2e-6, where="reset ") == []
    rng = np.random.default_rng(5)
    reached, off_route = 0, 0
    for t in range(90):
        acts = np.zeros((E, N), dtype=np.int8)
        acts[:, 2] = 3 if t % 9 < 5 else 0          # agent 2 drifts right, off its left-turn route
        acts[1] = np.where(rng.random(N) < 0.3, rng.integers(1, 4, N), 0)
        d, o = host(sim.step(torch.from_numpy(acts).cuda())), ob.step(acts)
        bad = parity.compare(d, o, tol64=1e-9, tol32=2e-5, where=f"t{t} ")
        assert bad == [], "\n".join(bad[:8])
        reached += int(d["events"][:, nat.EV_REACHED_GOAL].sum())
        off_route += int(d["events"][:, nat.EV_OFF_ROUTE].sum())
        parity.sync_oracle_from_device(ob, sim)
    sim.close()
    assert reached >= 2, reached   # goals 25 m ahead are reached within the rollout
    assert d["dist"].max() > 20.0


def test_missions_change_between_episodes_and_clear(nets, compiled_maps):
    """smx_set_missions replaces the table (knot lists of the old routes are dropped) and n = 0 clears it: the
    batch then equals one that never had missions."""
    import torch

    from smarts_amd.engine import BatchedSim, SimConfig

    cm = compiled_maps("4lane")
    missions = _missions_4lane(nets)
    N = len(missions)
    spawns = np.zeros((1, N, 4))
    for i, m in enumerate(missions):
        spawns[0, i] = (*m.spawn_pose(), 6.0)
    cfg = SimConfig(num_envs=1, num_vehicles=N, launch_strategy="large")
    a = BatchedSim(cm, cfg, spawns=spawns, missions=missions)
    b = BatchedSim(cm, cfg, spawns=spawns)
    acts = torch.zeros((1, N), dtype=torch.int8, device="cuda")
    a.reset(), b.reset()
    for _ in range(5):
        a.step(acts)
    a.set_missions(None)
    a.reset(), b.reset()
    for _ in range(12):
        oa, ob_ = a.step(acts), b.step(acts)
    torch.cuda.synchronize()
    for k in ob_:
        assert np.array_equal(oa[k].cpu().numpy(), ob_[k].cpu().numpy(), equal_nan=True), k
    a.close(), b.close()


def test_set_missions_rejects_bad_tables(compiled_maps):
    import ctypes as C

    from smarts_amd import _native as nat
    from smarts_amd.engine import BatchedSim, SimConfig, make_spawns
    from smarts_amd.missions import PlannedMission

    cm = compiled_maps("loop")
    sim = BatchedSim(cm, SimConfig(num_envs=1, num_vehicles=2), spawns=make_spawns(cm, 1, 2, episodes=1, seed=1))
    good = PlannedMission((0.0, 0.0), 0.0, (1.0, 2.0, 2.0), (cm.road_ids[0],))
    with pytest.raises(ValueError):
        sim.set_missions([good])  # one entry per slot
    recs = (nat.SmxMission * 2)()
    recs[0].route_off, recs[0].route_len, recs[0].goal_radius = 0, 1, 2.0
    for roads, n, msg in (([len(cm.road_ids)], 1, "road index"), ([0], 0, "route range")):
        arr = (C.c_int32 * 1)(*roads)
        rc = sim.lib.smx_set_missions(sim.handle, recs, 2, arr, n)
        assert rc != 0 and msg in sim.lib.smx_last_error(sim.handle).decode()
    recs[0].goal_radius = -1.0
    rc = sim.lib.smx_set_missions(sim.handle, recs, 2, (C.c_int32 * 1)(0), 1)
    assert rc != 0 and "PositionalGoal" in sim.lib.smx_last_error(sim.handle).decode()
    sim.set_missions([good, None])
    sim.close()


def test_hiway_env_mission_ends_at_its_goal():
    """gym surface: a Laner agent with a sstudio-style mission starts at the route's begin (front bumper on the
    start point), sees waypoints of its route only, and is done with events.reached_goal at the goal."""
    from smarts_amd.env import Agent, AgentInterface, AgentSpec, AgentType, HiWayEnv
    from smarts_amd.missions import Mission, Route

    spec = AgentSpec(interface=AgentInterface.from_type(AgentType.Laner, max_episode_steps=400),
                     agent_builder=lambda: Agent.from_function(lambda _: "keep_lane"))
    missions = {"A": Mission(Route(begin=("edge-west-WE", 1, 80), end=("edge-east-WE", 1, 25)))}
    env = HiWayEnv(scenarios=["scenarios/intersections/4lane"], agent_specs={"A": spec, "B": spec}, seed=3,
                   missions=missions)
    obs = env.reset()
    ego = obs["A"].ego_vehicle_state
    assert ego.mission.goal.position == (169.4, 68.4) and ego.mission.goal.radius == 2.0 and ego.mission.route_roads[-1] == "edge-east-WE"
    assert obs["B"].ego_vehicle_state.mission.goal == "EndlessGoal"  # no mission given: endless, empty route
    # Pose.from_front_bumper: the centre is half a chassis length behind the start point, eastbound
    assert np.allclose(ego.position[:2], (80.0 - 1.84, 68.4), atol=1e-9) and abs(float(ego.heading) + np.pi / 2) < 1e-6  # float32 row
    route_roads = {"edge-west-WE", ":junction-intersection_13", "edge-east-WE"}
    done, reached, ticks = {"__all__": False}, False, 0
    while not done["__all__"] and ticks < 400:
        if "A" in obs:
            roads = {wp.lane_id.rsplit("_", 1)[0] for path in obs["A"].waypoint_paths for wp in path}
            assert roads <= route_roads, roads
        obs, rew, done, info = env.step({a: "keep_lane" for a in obs})
        ticks += 1
        if done.get("A"):
            reached = obs["A"].events.reached_goal
            assert reached and not obs["A"].events.off_route
            # PositionalGoal(radius 2) 25 m down edge-east-WE lane 1 (scenario.py:687-691)
            assert np.hypot(obs["A"].ego_vehicle_state.position[0] - 169.4, obs["A"].ego_vehicle_state.position[1] - 68.4) <= 2.0
            break
    assert reached, ticks
    env.close()


@pytest.mark.parametrize("strategy", ["small", "large", "large_one_lane"])
def test_fixed_route_rollout_on_minicity(strategy, nets, compiled_maps):
    """Routes of the reference-generated fixture (several junctions each) driven on the big map: every output of
    every tick against the oracle, teacher-forced; one agent of three keeps an endless mission."""
    import torch

    from smarts_amd.engine import BatchedSim, SimConfig
    from smarts_amd.missions import Mission, Route, plan_mission

    cm, net = compiled_maps("minicity"), nets("minicity")
    g = np.load(os.path.join(GOLDEN, "missions_minicity.npz"))
    off = g["route_off"]
    missions = []
    for k in (1, 2):
        a, b = (str(x) for x in g["route_pairs"][k])
        m = plan_mission(net, Mission(Route(begin=(a, 0, 3.0), end=(b, 0, "max"))))
        assert list(m.route_roads) == [str(r) for r in g["route_roads"][off[k]:off[k + 1]]]  # the planner, end to end
        missions.append(m)
    missions.append(None)
    E, N = 2, 3
    spawns = np.zeros((1, E * N, 4))
    free = missions[0].spawn_pose()
    for e in range(E):
        for i, m in enumerate(missions):
            x, y, h = m.spawn_pose() if m is not None else (free[0] + 7.0 * np.sin(free[2]), free[1] - 7.0 * np.cos(free[2]), free[2])
            spawns[0, e * N + i] = (x, y, h, 7.0 + 2 * e)
    cfg = SimConfig(num_envs=E, num_vehicles=N, neighbors=True, nb_radius=50.0, launch_strategy=strategy)
    sim = BatchedSim(cm, cfg, spawns=spawns, missions=missions)
    ob = parity.OracleBatch(net, cm, cfg, spawns[0], missions=missions)

    def host(out):
        torch.cuda.synchronize()
        return {k: v.cpu().numpy().reshape((-1,) + tuple(v.shape[2:])) for k, v in out.items() if k != "env_done"}

    d, o = host(sim.reset()), ob.reset_observe()
    assert parity.compare(d, o, tol64=1e-9, tol32=2e-6, where="reset ") == []
    rng = np.random.default_rng(17)
    for t in range(70):
        acts = np.where(rng.random((E, N)) < 0.85, 0, rng.integers(1, 4, (E, N))).astype(np.int8)
        d, o = host(sim.step(torch.from_numpy(acts).cuda())), ob.step(acts)
        bad = parity.compare(d, o, tol64=1e-9, tol32=2e-5, where=f"t{t} ")
        assert bad == [], "\n".join(bad[:8])
        parity.sync_oracle_from_device(ob, sim)
    sim.close()
    assert d["dist"].max() > 30.0
