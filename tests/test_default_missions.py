"""The host restatement of the reference's default missions (smarts_amd/missions.py random_endless_missions) against
the reference's own draw (tests/golden/default_missions.npz, written by tests/golden/gen_golden.py from
smarts.core.plan.Mission.random_endless_mission over the reference's SumoRoadNetwork.random_route)."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "default_missions.npz")


@pytest.mark.parametrize("name", ["loop", "4lane", "minicity"])
@pytest.mark.parametrize("rolls", [0, 3])
def test_random_endless_missions_equal_the_references(name, rolls, nets):
    from smarts_amd.missions import random_endless_missions

    g = np.load(GOLDEN)
    ms = random_endless_missions(nets(name), 4, 42, scenario_rolls=rolls)
    pos = np.array([m.start_position for m in ms])
    head = np.array([m.start_heading for m in ms])
    assert np.array_equal(pos, g[f"{name}_rolls{rolls}_position"])  # bit for bit: same stream, same arithmetic
    assert np.array_equal(head, g[f"{name}_rolls{rolls}_heading"])
    assert all(m.route_roads == () for m in ms)  # endless: an empty route


def test_reference_spawn_table_shape_and_first_episode(nets):
    from smarts_amd.missions import random_endless_missions, reference_spawn_table

    t = reference_spawn_table(nets("loop"), 3, 4, 42, episodes=2)
    assert t.shape == (2, 12, 4) and (t[..., 3] == 0).all()
    for e in range(3):  # env e draws from the stream of seed + e (parallel_env.py:190-202)
        ms = random_endless_missions(nets("loop"), 4, 42 + e)
        assert np.allclose(t[0, e * 4:(e + 1) * 4, :3], [m.spawn_pose() for m in ms])
    assert not np.allclose(t[0], t[1])  # the next episode continues the stream
