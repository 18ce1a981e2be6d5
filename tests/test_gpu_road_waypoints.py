"""GPU parity of the RoadWaypointsSensor kernel (k_road_waypoints; reference sensors.py:991-1040).

* against the reference-generated fixture ``tests/golden/road_waypoints_<map>.npz`` directly: the sensor's lane
  list and order, the number of paths per lane, and every waypoint of the kept paths (lane ids / indices / counts
  exact, positions <= 1e-9, float32 rows to float32 rounding); poses where the reference itself raises (nearest
  lane junction-internal) are skipped;
* against the oracle on a teacher-forced rollout (every tick, with a fixed route on one agent).
"""
import os

import numpy as np
import pytest

import parity
import tie_sensitive
from conftest import GOLDEN
from test_gpu_golden import _host, _sim_at_poses

pytestmark = pytest.mark.gpu

MAP_NAMES = ["loop", "4lane", "minicity"]


@pytest.mark.parametrize("name", MAP_NAMES)
def test_road_waypoints_equal_the_reference(name, compiled_maps):
    from smarts_amd.missions import PlannedMission

    cm = compiled_maps(name)
    g = np.load(os.path.join(GOLDEN, f"road_waypoints_{name}.npz"))
    lane_no = np.array([cm.lane_ids.index(str(l)) for l in g["lane_ids"]])
    L, Q, H = 8, 6, int(g["horizon"])
    R = 2 * H + 1
    differing, answered = [], 0
    for routed in (0, 1):
        rows = np.flatnonzero((g["routed"] == routed) & (g["raised"] == 0))
        if len(rows) == 0:
            continue
        sim = _sim_at_poses(cm, g["poses"][rows], waypoints=False, road_waypoints=True, rw_horizon=H, rw_lanes=L, rw_paths=Q)
        if routed:
            sim.set_missions([PlannedMission((0.0, 0.0), 0.0, (1e7, 1e7, 1.0), tuple(str(r) for r in g["route_roads"]))])
        out = sim.reset()
        o = {k: _host(out[k])[:, 0] for k in out if k.startswith("rw_")}
        sim.close()
        for j, i in enumerate(rows):
            l0, l1 = g["lane_off"][i], g["lane_off"][i + 1]
            ok = o["rw_lane_count"][j] == l1 - l0
            ok = ok and np.array_equal(o["rw_lane"][j, :min(l1 - l0, L)], lane_no[g["lane"][l0:l1]][:L]) and (o["rw_lane"][j, l1 - l0:] == -1).all()
            for l in range(min(l1 - l0, L)):
                p0, p1 = g["path_off"][l0 + l], g["path_off"][l0 + l + 1]
                ok = ok and o["rw_path_count"][j, l] == p1 - p0
                for p in range(min(p1 - p0, Q)):
                    a, b = g["wp_off"][p0 + p], g["wp_off"][p0 + p + 1]
                    n = b - a
                    ok = ok and o["rw_count"][j, l, p] == n
                    if not ok:
                        break
                    sl = slice(a, b)
                    ok = ok and np.array_equal(o["rw_lane_id"][j, l, p, :n], lane_no[g["wp_lane"][sl]])
                    ok = ok and np.array_equal(o["rw_lane_index"][j, l, p, :n], g["lane_index"][sl])
                    ok = ok and np.abs(o["rw_pos"][j, l, p, :n, 0] - g["x"][sl]).max() <= 1e-9
                    ok = ok and np.abs(o["rw_pos"][j, l, p, :n, 1] - g["y"][sl]).max() <= 1e-9
                    for dev, ref in ((o["rw_heading"], g["heading"]), (o["rw_lane_width"], g["width"]), (o["rw_speed_limit"], g["speed"])):
                        ok = ok and np.abs(dev[j, l, p, :n] - ref[sl].astype(np.float32)).max() <= 2e-6 * max(1.0, np.abs(ref[sl]).max())
                ok = ok and (o["rw_count"][j, l, min(p1 - p0, Q):] == 0).all()
            if not ok:
                differing.append(int(i))
            answered += 1
    assert answered >= 25
    assert sorted(differing) == tie_sensitive.ROAD_WAYPOINTS[name], differing


@pytest.mark.parametrize("strategy", ["small", "large"])
def test_road_waypoints_rollout_against_the_oracle(strategy, nets, compiled_maps):
    import torch

    from smarts_amd.engine import BatchedSim, SimConfig
    from smarts_amd.missions import Mission, Route, plan_mission

    cm, net = compiled_maps("4lane"), nets("4lane")
    missions = [plan_mission(net, Mission(Route(begin=("edge-west-WE", 1, 60), end=("edge-east-WE", 1, 40)))), None, None]
    E, N = 2, 3
    starts = [missions[0].spawn_pose(), (125.2, 120.0, np.pi), (134.8, 20.0, 0.0)]  # west->east, southbound, northbound
    spawns = np.zeros((1, E * N, 4))
    for e in range(E):
        for i, (x, y, h) in enumerate(starts):
            spawns[0, e * N + i] = (x, y, h, 9.0 + e)
    cfg = SimConfig(num_envs=E, num_vehicles=N, road_waypoints=True, rw_horizon=20, rw_lanes=6, rw_paths=3, launch_strategy=strategy)
    sim = BatchedSim(cm, cfg, spawns=spawns, missions=missions)
    ob = parity.OracleBatch(net, cm, cfg, spawns[0], missions=missions)

    def host(out):
        torch.cuda.synchronize()
        return {k: v.cpu().numpy().reshape((-1,) + tuple(v.shape[2:])) for k, v in out.items() if k != "env_done"}

    d, o = host(sim.reset()), ob.reset_observe()
    assert parity.compare(d, o, tol64=1e-9, tol32=2e-6, where="reset ") == []
    seen_lanes = 0
    for t in range(40):
        acts = np.zeros((E, N), dtype=np.int8)
        d, o = host(sim.step(torch.from_numpy(acts).cuda())), ob.step(acts)
        bad = parity.compare(d, o, tol64=1e-9, tol32=2e-5, where=f"t{t} ")
        assert bad == [], "\n".join(bad[:8])
        seen_lanes = max(seen_lanes, int(d["rw_lane_count"].max()))
        parity.sync_oracle_from_device(ob, sim)
    sim.close()
    assert seen_lanes >= 4  # the oncoming road's lanes are reported beside the ego road's


def test_hiway_env_road_waypoints_observation():
    """gym surface: AgentInterface(road_waypoints=RoadWaypoints(horizon)) -> Observation.road_waypoints.lanes keyed by
    lane id, every path starting `horizon` behind the vehicle and 2 x horizon + 1 waypoints long where the road goes on."""
    from smarts_amd.env import Agent, AgentInterface, AgentSpec, AgentType, HiWayEnv
    from smarts_amd.env.agent_interface import RoadWaypoints

    itf = AgentInterface.from_type(AgentType.Laner, road_waypoints=RoadWaypoints(horizon=16), max_episode_steps=50)
    spec = AgentSpec(interface=itf, agent_builder=lambda: Agent.from_function(lambda _: "keep_lane"))
    env = HiWayEnv(scenarios=["scenarios/loop"], agent_specs={"A": spec}, seed=11)
    obs = env.reset()
    for _ in range(3):
        rw = obs["A"].road_waypoints
        ego = obs["A"].ego_vehicle_state
        assert ego.lane_id in rw.lanes and len(rw.lanes) >= 2  # the ego road's lanes
        for lane_id, paths in rw.lanes.items():
            assert len(paths) >= 1
            for path in paths:
                assert 1 <= len(path) <= 33
                # the paths reach back behind the vehicle: the first waypoint lies ~horizon metres from it
                d0 = np.linalg.norm(path[0].pos - ego.position[:2])
                assert d0 > 8.0, (lane_id, d0)
        obs, _, _, _ = env.step({"A": "keep_lane"})
    env.close()
