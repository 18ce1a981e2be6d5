"""The reference's behavioural envelopes for the closed loop (controller + vehicle dynamics), run on
the oracle.  pybullet's multibody solve is substituted by a planar single-track model (DESIGN.md
§4): no golden poses exist to pin it, but the reference's own tests bound how the closed loop must
behave, and the substituted model has to stay inside those bounds."""
import math

import numpy as np
import pytest

import parity
from oracle import controller as ctl
from oracle.dynamics import VehicleBody


def test_lane_following_envelope(nets, compiled_maps):
    """test_controller_lane.py:99-147: a Laner agent sending keep_lane on scenarios/loop for 500 ticks —
    speed never falls back under 5 km/h once it exceeded it, mean speed > 5 m/s, lateral error to
    the first waypoint of the current lane < 2.2 m at every tick and < 1 m on average."""
    from smarts_amd.engine import SimConfig, make_spawns

    cm = compiled_maps("loop")
    cfg = SimConfig(num_envs=1, num_vehicles=1, done_off_road=False, done_off_route=False, done_collision=False)
    spawns = make_spawns(cm, 1, 1, episodes=1, seed=42)
    ob = parity.OracleBatch(nets("loop"), cm, cfg, spawns[0])
    env = ob.envs[0]
    obs = env.reset_observe()
    detected, speeds, lateral = False, [], []
    for _ in range(500):
        o = obs[0]
        speed = o["ego"]["speed"]
        detected = detected or speed > 5 / 3.6
        if detected:
            speeds.append(speed)
        paths = o["waypoint_paths"]
        pos = o["ego"]["position"]
        current = ctl.find_current_lane(paths, pos)
        wp = paths[current][0]
        lateral.append(abs(wp.signed_lateral_error(pos[:2])))
        obs, _, dones = env.step(["keep_lane"])
        assert not dones[0]
    assert min(speeds) > 5 / 3.6
    assert sum(speeds) / len(speeds) > 5
    assert max(lateral) < 2.2
    assert sum(lateral) / len(lateral) < 1


@pytest.mark.parametrize("radius", [10, 20])
@pytest.mark.parametrize("omega", [0.1, 0.15, 0.2])
def test_trajectory_tracking_envelope(radius, omega):
    """test_trajectory_controller.py:104-167: PD tracking of a circular arc (15-point trajectories,
    speed R * omega) for half of the half circle; the final position error stays within 10 m."""
    dt = 0.1  # test_trajectory_controller.py:44, 64-72: fixedTimeStep 0.1 s in 24 substeps, like SMARTS itself

    def build_trajectory(step_num):
        n = 15
        return [
            [-(radius - radius * math.cos((step_num + i) * omega * dt)) for i in range(n)],
            [radius * math.sin((step_num + i) * omega * dt) for i in range(n)],
            [(step_num + i) * omega * dt for i in range(n)],
            [radius * omega for _ in range(n)],
        ]

    body = VehicleBody(0.0, 0.0, 0.0, 0.0)
    state = ctl.TrajectoryTrackingControllerState()
    n_steps = int(0.5 * 3.14 / (omega * dt))
    traj = None
    for step_num in range(n_steps):
        traj = build_trajectory(step_num)
        thr, brk, steer = ctl.perform_trajectory_tracking_pd(traj, body, state, dt)
        body.control(throttle=thr, brake=brk, steering=steer)
        body.step(dt)
    err = math.hypot(body.x - traj[0][0], body.y - traj[1][0])
    assert np.isfinite(err) and err <= 10
