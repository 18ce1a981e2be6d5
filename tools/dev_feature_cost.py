"""Developer: what the optional rows cost per tick on one MI355X — fixed-route missions on C3's shape
(4lane 2048 x 16) and the road-waypoints sensor on C2's shape (loop 1024 x 8) — against the same batch without
them.  HIP-event timing of smx_step (smx_set_timing(1)), median over the sampled ticks.
    python tools/dev_feature_cost.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from smarts_amd.engine import BatchedSim, SimConfig, make_spawns  # noqa: E402
from smarts_amd.map_compiler import compile_map  # noqa: E402
from smarts_amd.missions import Mission, Route, plan_mission  # noqa: E402
from smarts_amd.sumo_map import load_net  # noqa: E402


def tick_ms(sim, E, N, steps=150, warm=30):
    actions = torch.from_numpy(bench.action_stream(E, N, 42, 0)).cuda()
    sim.reset()
    for i in range(warm):
        sim.step(actions[i % bench.ACTION_CYCLE])
    sim.set_timing(1)
    for i in range(steps):
        sim.step(actions[(warm + i) % bench.ACTION_CYCLE])
    torch.cuda.synchronize()
    ms = np.asarray(sim.read_step_ms())
    alive = float((sim.flags & 1).float().mean())
    sim.set_timing(0)
    return float(np.median(ms)), alive


def main():
    out = {}
    # ---- missions on 4lane 2048 x 16: every slot gets one of the junction's routes
    net = load_net(os.path.join(ROOT, "smarts_amd", "scenarios", "intersections", "4lane"))
    cm = compile_map(net)
    E, N = 2048, 16
    arms = [("edge-west-WE", "edge-east-WE"), ("edge-north-NS", "edge-east-WE"), ("edge-south-SN", "edge-west-EW"),
            ("edge-east-EW", "edge-west-EW"), ("edge-west-WE", "edge-south-NS"), ("edge-north-NS", "edge-south-NS"),
            ("edge-south-SN", "edge-north-SN"), ("edge-east-EW", "edge-north-SN")]
    missions = []
    for s in range(N):
        a, b = arms[s % len(arms)]
        missions.append(plan_mission(net, Mission(Route(begin=(a, s % 2, 6.0 + 9.0 * (s // len(arms))), end=(b, s % 2, "max")))))
    spawns = np.zeros((1, E * N, 4))
    for e in range(E):
        for s, m in enumerate(missions):
            spawns[0, e * N + s] = (*m.spawn_pose(), 8.0)
    kw = dict(num_envs=E, num_vehicles=N, neighbors=True, nb_radius=50.0, auto_reset=True)
    for label, ms_ in (("endless", None), ("fixed_routes", missions)):
        sim = BatchedSim(cm, SimConfig(**kw), spawns=spawns, missions=ms_)
        out[f"4lane_2048x16_{label}"] = tick_ms(sim, E, N)
        sim.close()
    # ---- road waypoints on loop 1024 x 8
    cm = compile_map(load_net(os.path.join(ROOT, "smarts_amd", "scenarios", "loop")))
    E, N = 1024, 8
    spawns = make_spawns(cm, E, N, episodes=4, seed=42)
    kw = dict(num_envs=E, num_vehicles=N, neighbors=True, nb_radius=50.0, auto_reset=True)
    for label, extra in (("off", {}), ("road_waypoints_h32_8x4", dict(road_waypoints=True, rw_horizon=32, rw_lanes=8, rw_paths=4))):
        sim = BatchedSim(cm, SimConfig(**kw, **extra), spawns=spawns)
        out[f"loop_1024x8_{label}"] = tick_ms(sim, E, N)
        sim.close()
    for k, (ms, alive) in out.items():
        print(f"{k}: {ms:.4f} ms/tick (alive fraction at the end {alive:.2f})")


if __name__ == "__main__":
    main()
