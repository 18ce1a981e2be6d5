"""GPU parity: the HIP path (through the C-ABI) against the oracle on the same seeded inputs.

Bars (BASELINE.json north_star): lane ids / lane indices / counts / event flags / done bit-exact;
vehicle pose within 1e-5 abs.  Teacher-forced ticks (oracle state re-synchronised from the
device each tick) are held to much tighter float tolerances; a free-running rollout is held to
the 1e-5 bar over a window in which libm last-ulp differences cannot grow past it.
"""
import numpy as np
import pytest

import parity

pytestmark = pytest.mark.gpu


def _host(out):
    import torch

    torch.cuda.synchronize()
    return {k: v.cpu().numpy().reshape((-1,) + tuple(v.shape[2:])) for k, v in out.items() if k != "env_done"}


def _actions(rng, E, N):
    # keep_lane with probability 0.8, else uniform over the other three
    return np.where(rng.random((E, N)) < 0.8, 0, rng.integers(1, 4, (E, N))).astype(np.int8)


def _make(name, E, N, nets, compiled_maps, seed, **cfg_kw):
    from smarts_amd.engine import BatchedSim, SimConfig, make_spawns

    cm = compiled_maps(name)
    cfg = SimConfig(num_envs=E, num_vehicles=N, neighbors=True, nb_radius=50.0, **cfg_kw)
    spawns = make_spawns(cm, E, N, episodes=2, seed=seed)
    sim = BatchedSim(cm, cfg, spawns=spawns)
    ob = parity.OracleBatch(nets(name), cm, cfg, spawns[0])
    return sim, ob, cfg


@pytest.mark.parametrize("name,E,N,T,seed", [("loop", 8, 8, 80, 11), ("4lane", 4, 16, 60, 12), ("minicity", 2, 16, 40, 13),
                                              ("loop", 2, 32, 30, 14)])
def test_teacher_forced_ticks(name, E, N, T, seed, nets, compiled_maps):
    import torch

    sim, ob, cfg = _make(name, E, N, nets, compiled_maps, seed)
    d, o = _host(sim.reset()), ob.reset_observe()
    assert parity.compare(d, o, tol64=1e-9, tol32=2e-6, where="reset ") == []
    rng = np.random.default_rng(seed)
    for t in range(T):
        acts = _actions(rng, E, N)
        if t % 7 == 3:
            acts[0, 0] = -1  # an agent that sends no action this tick
        d, o = _host(sim.step(torch.from_numpy(acts).cuda())), ob.step(acts)
        bad = parity.compare(d, o, tol64=1e-9, tol32=2e-5, where=f"{name} t{t} ")
        assert bad == [], "\n".join(bad[:8])
        parity.sync_oracle_from_device(ob, sim)
    sim.close()


@pytest.mark.parametrize("name,E,N,T,seed,sensor", [("loop", 2, 32, 20, 31, "ogm"), ("minicity", 2, 32, 20, 32, "lidar"),
                                                     ("4lane", 2, 16, 12, 33, "basic_lidar+ogm"),
                                                     ("loop", 1, 8, 4, 34, "default_ogm")])
def test_ogm_and_lidar_sensors(name, E, N, T, seed, sensor, nets, compiled_maps):
    """BASELINE.json configs[3] (OGM 64 x 64 over 50 m) and configs[4] (100-ray planar lidar) at
    oracle-sized batches: grids and hit flags bit-exact, hit points to 1e-9."""
    import torch

    from smarts_amd.lidar import BasicLidar, Planar100

    kw = {}
    if sensor == "default_ogm":
        kw.update(ogm=True)  # agent_interface.py:42-51: 256 x 256 @ 50/256 — a 64 KiB tile, its own launch
    elif "ogm" in sensor:
        kw.update(ogm=True, ogm_width=64, ogm_height=64, ogm_resolution=50 / 64)
    if "lidar" in sensor:
        kw.update(lidar=BasicLidar if "basic" in sensor else Planar100)
    sim, ob, cfg = _make(name, E, N, nets, compiled_maps, seed, **kw)
    d, o = _host(sim.reset()), ob.reset_observe()
    assert parity.compare(d, o, tol64=1e-9, tol32=2e-6, where="reset ") == []
    if "ogm" in sensor:
        side = cfg.ogm_width
        g = d["ogm"].reshape(E * N, side, side)
        assert g[:, side // 2 - 1:side // 2 + 1, side // 2 - 1:side // 2 + 1].min() == 255  # own footprint at the centre
    rng = np.random.default_rng(seed)
    seen_hits = 0
    for t in range(T):
        acts = _actions(rng, E, N)
        d, o = _host(sim.step(torch.from_numpy(acts).cuda())), ob.step(acts)
        bad = parity.compare(d, o, tol64=1e-9, tol32=2e-5, where=f"{name} {sensor} t{t} ")
        assert bad == [], "\n".join(bad[:8])
        if "lidar" in sensor:
            seen_hits += int(d["lidar_hit"].sum())
        parity.sync_oracle_from_device(ob, sim)
    if "lidar" in sensor:
        assert seen_hits > 0
    sim.close()


@pytest.mark.parametrize("name,E,N,T,seed,grid", [("loop", 2, 8, 8, 35, (64, 64, 50 / 64)), ("4lane", 1, 12, 6, 36, (64, 32, 0.5)),
                                                   ("minicity", 1, 6, 3, 37, None)])
def test_drivable_area_grid_map(name, E, N, T, seed, grid, nets, compiled_maps):
    """DrivableAreaGridMap (sensors.py:675-716; defaults 256 x 256 @ 50/256, agent_interface.py:29-38)
    bit-exact against the oracle's raster, and the reference's own check (test_observations.py:150-153):
    vehicles on the road sit on drivable pixels."""
    import torch

    from smarts_amd import _native as nat

    kw = dict(dagm=True)
    if grid is not None:
        kw.update(dagm_width=grid[0], dagm_height=grid[1], dagm_resolution=grid[2])
    sim, ob, cfg = _make(name, E, N, nets, compiled_maps, seed, **kw)
    d, o = _host(sim.reset()), ob.reset_observe()
    assert parity.compare(d, o, tol64=1e-9, tol32=2e-6, where="reset ") == []
    rng = np.random.default_rng(seed)
    for t in range(T):
        g = d["dagm"].reshape(E * N, cfg.dagm_height, cfg.dagm_width)
        on_road = (d["active"].reshape(-1) != 0) & (d["events"][:, nat.EV_OFF_ROAD] == 0)
        cy, cx = cfg.dagm_height // 2, cfg.dagm_width // 2
        assert on_road.any() and (g[on_road, cy - 2:cy + 2, cx - 2:cx + 2].max(axis=(1, 2)) == 255).all()
        assert 0 < (g[on_road] == 255).mean() < 1  # some road, not everything
        acts = _actions(rng, E, N)
        d, o = _host(sim.step(torch.from_numpy(acts).cuda())), ob.step(acts)
        bad = parity.compare(d, o, tol64=1e-9, tol32=2e-5, where=f"{name} dagm t{t} ")
        assert bad == [], "\n".join(bad[:8])
        parity.sync_oracle_from_device(ob, sim)
    sim.close()


@pytest.mark.parametrize("space,name,E,N,T,seed", [("Continuous", "loop", 4, 8, 40, 41), ("ActuatorDynamic", "4lane", 2, 16, 30, 42),
                                                  ("LaneWithContinuousSpeed", "loop", 4, 8, 50, 43),
                                                  ("LaneWithContinuousSpeed", "minicity", 2, 16, 25, 44)])
def test_float_action_spaces(space, name, E, N, T, seed, nets, compiled_maps):
    """Controllers.perform_action for Continuous / ActuatorDynamic / LaneWithContinuousSpeed
    (controllers/__init__.py:94-124), teacher-forced against the oracle; target speeds include the
    clip window of the heading gain (2.02-2.06 m/s) and 0."""
    import torch

    sim, ob, cfg = _make(name, E, N, nets, compiled_maps, seed, action_space=space)
    d, o = _host(sim.reset()), ob.reset_observe()
    assert parity.compare(d, o, tol64=1e-9, tol32=2e-6, where="reset ") == []
    rng = np.random.default_rng(seed)
    for t in range(T):
        if space == "LaneWithContinuousSpeed":
            speed = rng.choice([0.0, 2.03, 2.045, 5.0, 9.5, 14.0, 18.0], size=(E, N))
            change = rng.choice([0.0, 0.0, 0.0, 1.0, -1.0], size=(E, N))
            acts = np.stack([speed, change, np.zeros((E, N))], axis=-1).astype(np.float32)
        else:
            acts = np.stack([rng.uniform(-0.2, 1.2, (E, N)), np.where(rng.random((E, N)) < 0.2, rng.uniform(0, 1, (E, N)), 0.0),
                             rng.uniform(-1.3, 1.3, (E, N)) * 0.3], axis=-1).astype(np.float32)
        if t % 5 == 2:
            acts[0, 0, 0] = np.nan  # an agent that sends no action this tick
        d, o = _host(sim.step(torch.from_numpy(acts).cuda())), ob.step(acts.astype(np.float64))
        bad = parity.compare(d, o, tol64=1e-9, tol32=2e-5, where=f"{space} {name} t{t} ")
        assert bad == [], "\n".join(bad[:8])
        parity.sync_oracle_from_device(ob, sim)
    sim.close()


VARIANTS = {
    "no_waypoints_sensor": dict(waypoints=False),                       # trip meter from the lookahead-1 query (sensors.py:270-275)
    "no_accelerometer_unlimited_radius": dict(accelerometer=False, nb_radius=None),
    "short_lookahead_two_paths": dict(wp_lookahead=8, wp_paths=2, wp_len=9),
    "long_rows": dict(wp_lookahead=32, wp_paths=6, wp_len=33),
    "all_done_criteria": dict(done_on_shoulder=True, done_wrong_way=True, done_not_moving=True, not_moving_time=0.5,
                              not_moving_distance=1.0, done_collision=False, max_episode_steps=15),
    "half_timestep": dict(dt=0.05),
    # DoneCriteria.agents_alive: the fleet thins out by max_episode_steps on a stagger-free clock, so
    # force early exits with collisions off and a shoulder criterion; the rest follow by the rule
    "agents_alive": dict(done_on_shoulder=True, alive_min_ego="N", alive_lists=(((0, 1, 2), 2),), max_episode_steps=14),
    "no_neighbours": dict(neighbors=False),
}


@pytest.mark.parametrize("variant", sorted(VARIANTS))
@pytest.mark.parametrize("name,E,N", [("loop", 3, 7), ("minicity", 1, 64)])
def test_config_variants(variant, name, E, N, nets, compiled_maps):
    """Sensor / done-criteria / shape options of AgentInterface (agent_interface.py:214-297) and odd
    batch shapes (E*N not a multiple of the workgroup, the 64-vehicle limit), teacher-forced."""
    import torch

    from smarts_amd.engine import BatchedSim, SimConfig, make_spawns

    cm = compiled_maps(name)
    kw = dict(num_envs=E, num_vehicles=N, neighbors=True, nb_radius=50.0)
    kw.update(VARIANTS[variant])
    if kw.get("alive_min_ego") == "N":
        kw["alive_min_ego"] = N  # the first agent to leave takes the others with it one tick later
    cfg = SimConfig(**kw)
    spawns = make_spawns(cm, E, N, episodes=2, seed=77)
    sim = BatchedSim(cm, cfg, spawns=spawns)
    ob = parity.OracleBatch(nets(name), cm, cfg, spawns[0])
    d, o = _host(sim.reset()), ob.reset_observe()
    assert parity.compare(d, o, tol64=1e-9, tol32=2e-6, where="reset ") == []
    rng = np.random.default_rng(77)
    T = 8 if N == 64 else 18
    seen_alive_done = 0
    for t in range(T):
        acts = _actions(rng, E, N)
        if variant == "all_done_criteria":
            acts[:, ::2] = 1  # slow_down: half of the fleet stops and trips not_moving
        d, o = _host(sim.step(torch.from_numpy(acts).cuda())), ob.step(acts)
        bad = parity.compare(d, o, tol64=1e-9, tol32=2e-5, where=f"{variant} {name} t{t} ")
        assert bad == [], "\n".join(bad[:8])
        seen_alive_done += int(d["events"][:, 8].sum())
        parity.sync_oracle_from_device(ob, sim)
    if variant == "agents_alive":
        assert seen_alive_done > 0  # the criterion fired
    sim.close()


@pytest.mark.parametrize("name,E,agents,social,T,seed", [("loop", 3, 6, 10, 60, 51), ("4lane", 2, 4, 12, 40, 52),
                                                         ("minicity", 1, 8, 40, 25, 53)])
def test_scripted_social_traffic(name, E, agents, social, T, seed, nets, compiled_maps):
    """include/smx.h smx_config.num_social: kinematic lane followers in the last slots of every env —
    visible to the agents' neighbourhood / collision / OGM sensors, silent themselves."""
    import torch

    from smarts_amd.engine import BatchedSim, SimConfig, make_spawns

    cm = compiled_maps(name)
    N = agents + social
    cfg = SimConfig(num_envs=E, num_vehicles=N, num_social=social, neighbors=True, nb_radius=60.0, done_collision=True,
                    ogm=(name == "loop"), ogm_width=64, ogm_height=64, ogm_resolution=50 / 64)
    spawns, where = make_spawns(cm, E, N, episodes=2, seed=seed, return_lanes=True)
    sim = BatchedSim(cm, cfg, spawns=spawns, social_spawns=where)
    ob = parity.OracleBatch(nets(name), cm, cfg, spawns[0], where[0])
    # the two lane graphs must list successors in the same order for the social model to agree
    for lane_idx in range(cm.n_lanes):
        outs = [cm.lane_ids[j] for j in cm.lane_out_idx[cm.lane_out_off[lane_idx]:cm.lane_out_off[lane_idx + 1]]]
        assert outs == [l.lane_id for l in ob.road_map.lane_by_id(cm.lane_ids[lane_idx]).outgoing_lanes]
    d, o = _host(sim.reset()), ob.reset_observe()
    assert parity.compare(d, o, tol64=1e-9, tol32=2e-6, where="reset ") == []
    assert d["active"].reshape(E, N)[:, agents:].sum() == 0  # social slots never observe
    rng = np.random.default_rng(seed)
    saw_social_neighbour = collided = 0
    pos0 = sim.state[0:2, :, agents:].clone()
    for t in range(T):
        acts = _actions(rng, E, N)
        d, o = _host(sim.step(torch.from_numpy(acts).cuda())), ob.step(acts)
        bad = parity.compare(d, o, tol64=1e-9, tol32=2e-5, where=f"{name} t{t} ")
        assert bad == [], "\n".join(bad[:8])
        saw_social_neighbour += int((d["nb_slot"] >= agents).sum())
        collided += int(d["events"][:, 0].sum())
        parity.sync_oracle_from_device(ob, sim)
    assert saw_social_neighbour > 0
    moved = (sim.state[0:2, :, agents:] - pos0).norm(dim=0)
    assert float(moved.median()) > 5.0  # the social fleet drove on (dead-end lanes stop their vehicles)
    sim.close()


@pytest.mark.parametrize("name,E,agents,social,T,seed", [("loop", 3, 4, 20, 80, 54), ("4lane", 2, 4, 12, 40, 55)])
def test_social_traffic_with_car_following(name, E, agents, social, T, seed, nets, compiled_maps):
    """include/smx.h SMX_SOCIAL_IDM (k_social): followers read leaders (agents included) at the start of
    the tick; device vs oracle, and the fleet keeps its distance where the constant-speed model would not."""
    import torch

    from smarts_amd.engine import BatchedSim, SimConfig, make_spawns

    cm = compiled_maps(name)
    N = agents + social
    cfg = SimConfig(num_envs=E, num_vehicles=N, num_social=social, social_model="idm", social_speed_factor=1.0,
                    neighbors=True, nb_radius=60.0, done_collision=False)
    spawns, where = make_spawns(cm, E, N, episodes=2, seed=seed, return_lanes=True)
    sim = BatchedSim(cm, cfg, spawns=spawns, social_spawns=where)
    ob = parity.OracleBatch(nets(name), cm, cfg, spawns[0], where[0])
    d, o = _host(sim.reset()), ob.reset_observe()
    assert parity.compare(d, o, tol64=1e-9, tol32=2e-6, where="reset ") == []
    rng = np.random.default_rng(seed)
    slowed = 0
    for t in range(T):
        acts = _actions(rng, E, N)
        acts[:, 0] = 1  # agent 0 of every env brakes to a stop: a standing obstacle for whoever follows it
        d, o = _host(sim.step(torch.from_numpy(acts).cuda())), ob.step(acts)
        bad = parity.compare(d, o, tol64=1e-9, tol32=2e-5, where=f"{name} idm t{t} ")
        assert bad == [], "\n".join(bad[:8])
        parity.sync_oracle_from_device(ob, sim)
        S = sim.state
        for e, env in enumerate(ob.envs):  # the device state the sync just copied is the oracle's own
            for k, sv in enumerate(env.social):
                assert abs(float(S[3, e, agents + k]) - sv.body.u) < 1e-9
        u = S[3, :, agents:]
        limit = torch.tensor([[cm.lane_speed[int(S[12, e, agents + k])] for k in range(social)] for e in range(E)], device=u.device)
        slowed += int((u < 0.7 * limit).sum())
    assert slowed > 0  # somebody was held up by a leader
    sim.close()


@pytest.mark.parametrize("name,E,N,T,seed", [("loop", 3, 6, 40, 61), ("minicity", 2, 12, 25, 62)])
def test_trajectory_action_space(name, E, N, T, seed, nets, compiled_maps):
    """ActionSpaceType.Trajectory (PD tracking, trajectory_tracking_controller.py:176-331): every
    agent tracks one of its own waypoint paths with a speed profile; teacher-forced vs the oracle."""
    import torch

    from oracle import controller as octl
    from smarts_amd.engine import pack_trajectory

    sim, ob, cfg = _make(name, E, N, nets, compiled_maps, seed, action_space="Trajectory")
    d, o = _host(sim.reset()), ob.reset_observe()
    assert parity.compare(d, o, tol64=1e-9, tol32=2e-6, where="reset ") == []
    rng = np.random.default_rng(seed)
    for t in range(T):
        packed = np.zeros((E, N, 4, 11))
        counts = np.zeros((E, N), dtype=np.int32)
        oracle_actions = [[([], [], [], []) for _ in range(N)] for _ in range(E)]
        wp_pos = d["wp_pos"].reshape(E, N, 4, 20, 3)
        wp_h = d["wp_heading"].reshape(E, N, 4, 20)
        wp_c = d["wp_count"].reshape(E, N, 5)
        act = d["active"].reshape(E, N)
        for e in range(E):
            for i in range(N):
                if not act[e, i] or wp_c[e, i, 0] == 0 or (t % 6 == 4 and i == 0):
                    continue  # gone, nothing to track, or a tick without an action
                p = int(rng.integers(min(int(wp_c[e, i, 0]), 4)))
                n = int(rng.choice([3, 7, 10, 11, 20]))
                n = min(n, int(wp_c[e, i, 1 + p]))
                v0 = float(rng.choice([0.0, 6.0, 11.0, 16.0, 22.0]))
                traj = (wp_pos[e, i, p, :n, 0].tolist(), wp_pos[e, i, p, :n, 1].tolist(),
                        [float(x) for x in wp_h[e, i, p, :n]], [v0 + 0.1 * k for k in range(n)])
                packed[e, i], counts[e, i] = pack_trajectory(traj)
                oracle_actions[e][i] = octl.unpack_trajectory(packed[e, i], n)
        d = _host(sim.step_trajectory(torch.from_numpy(packed), torch.from_numpy(counts)))
        parts = []
        for e, env in enumerate(ob.envs):
            obs, rew, dones = env.step(oracle_actions[e])
            parts.append(parity.pack(cfg, ob.lane_no, N, obs, rew, dones))
        o = ob._stack(parts)
        bad = parity.compare(d, o, tol64=1e-9, tol32=2e-5, where=f"trajectory {name} t{t} ")
        assert bad == [], "\n".join(bad[:8])
        parity.sync_oracle_from_device(ob, sim)
    sim.close()


def test_via_sensor(nets, compiled_maps):
    """ViaSensor (sensors.py:1090-1149) on scenarios/intersections/4lane with the via points of that
    scenario's mission; agents start on the approach lanes so that some vias get hit."""
    import torch

    from smarts_amd.engine import BatchedSim, SimConfig, make_spawns
    from smarts_amd.vias import Via, resolve_vias

    cm = compiled_maps("4lane")
    vias = resolve_vias(cm, [Via("edge-south-SN", 1, 30, 4), Via("edge-west-EW", 0, 20, 8), Via("edge-west-EW", 1, 50, 2),
                             Via("edge-west-EW", 0, 55, 5), Via("edge-south-SN", 0, 25, 13, hit_distance=3.0),
                             Via("edge-south-SN", 1, 45, 13, hit_distance=3.0)])
    E, N = 3, 8
    cfg = SimConfig(num_envs=E, num_vehicles=N, neighbors=True, nb_radius=50.0, via_max=4, done_off_route=False)
    spawns = make_spawns(cm, E, N, episodes=2, seed=71)
    # slots 0 and 1 start on the approach lanes, below the 13 m/s vias, at 13 m/s
    from smarts_amd.engine import lane_heading
    from smarts_amd.vias import _position_at_shape_offset

    for slot, (lane_id, off) in enumerate([("edge-south-SN_0", 6.0), ("edge-south-SN_1", 20.0)]):
        shape = cm.lane_shape(cm.lane_ids.index(lane_id))
        x, y = _position_at_shape_offset(shape, off)
        spawns[:, slot::N] = (x, y, lane_heading(shape, 0), 13.0)
    per_slot = [vias if i % 2 == 0 else vias[:2] + vias[4:] for i in range(N - 1)] + [[]]
    sim = BatchedSim(cm, cfg, spawns=spawns, vias=per_slot)
    ob = parity.OracleBatch(nets("4lane"), cm, cfg, spawns[0], vias=per_slot)
    d, o = _host(sim.reset()), ob.reset_observe()
    assert parity.compare(d, o, tol64=1e-9, tol32=2e-6, where="reset ") == []
    rng = np.random.default_rng(71)
    hits = near_rows = 0
    for t in range(45):
        acts = _actions(rng, E, N)
        d, o = _host(sim.step(torch.from_numpy(acts).cuda())), ob.step(acts)
        bad = parity.compare(d, o, tol64=1e-9, tol32=2e-5, where=f"vias t{t} ")
        assert bad == [], "\n".join(bad[:8])
        hits += int(sum(bin(int(x)).count("1") for x in d["via_hit"]))
        near_rows += int((d["via_near_count"] > 4).sum())
        parity.sync_oracle_from_device(ob, sim)
    assert hits > 0 and near_rows > 0  # vias were hit, and lists longer than the kept window occurred
    sim.close()


def test_collision_thresholds_of_the_reference(nets, compiled_maps):
    """test_collision.py:206-282 through the C-ABI: standing passenger boxes packed around a
    standing ego, 0.0501 m clear on every side -> no collision event; 0.0499 m -> every pair that
    touches reports it (and is done: DoneCriteria.collision)."""
    import torch

    from smarts_amd import _native as nat
    from smarts_amd.engine import BatchedSim, SimConfig, lane_heading, make_spawns
    from smarts_amd.vias import _position_at_shape_offset

    cm = compiled_maps("4lane")
    E, N = 2, 5
    cfg = SimConfig(num_envs=E, num_vehicles=N, neighbors=True, nb_radius=50.0)
    spawns = make_spawns(cm, E, N, episodes=1, seed=5)
    shape = cm.lane_shape(cm.lane_ids.index("edge-south-SN_0"))
    cx, cy = _position_at_shape_offset(shape, 40.0)
    h = lane_heading(shape, 0)
    f, r = np.array([-np.sin(h), np.cos(h)]), np.array([np.cos(h), np.sin(h)])
    for e, sep in enumerate((0.0501, 0.0499)):
        ring = [(0.0, 0.0), (3.68 + sep, 0.0), (0.0, 1.47 + sep), (-(3.68 + sep), 0.0), (0.0, -(1.47 + sep))]
        for k, (along, across) in enumerate(ring):
            x, y = np.array([cx, cy]) + along * f + across * r
            spawns[:, e * N + k] = (x, y, h, 0.0)
    sim = BatchedSim(cm, cfg, spawns=spawns)
    ob = parity.OracleBatch(nets("4lane"), cm, cfg, spawns[0])
    d, o = _host(sim.reset()), ob.reset_observe()
    assert parity.compare(d, o, tol64=1e-9, tol32=2e-6, where="reset ") == []
    acts = np.full((E, N), nat.ACTION_SLOW_DOWN, dtype=np.int8)  # target speed 0: nobody moves
    d, o = _host(sim.step(torch.from_numpy(acts).cuda())), ob.step(acts)
    assert parity.compare(d, o, tol64=1e-9, tol32=2e-5, where="t0 ") == []
    hit = d["events"][:, nat.EV_COLLISIONS].reshape(E, N)
    assert not hit[0].any() and hit[1].all()
    assert not d["done"].reshape(E, N)[0].any() and d["done"].reshape(E, N)[1].all()
    sim.close()


def test_free_running_rollout_pose_bar(nets, compiled_maps):
    import torch

    E, N = 8, 8
    sim, ob, cfg = _make("loop", E, N, nets, compiled_maps, 21)
    sim.reset()
    ob.reset_observe()
    rng = np.random.default_rng(21)
    for t in range(30):
        acts = _actions(rng, E, N)
        d, o = _host(sim.step(torch.from_numpy(acts).cuda())), ob.step(acts)
        bad = parity.compare(d, o, tol64=1e-5, tol32=1e-3, where=f"t{t} ")
        assert bad == [], "\n".join(bad[:8])
    sim.close()


@pytest.mark.parametrize("name,E,N", [("loop", 4, 8), ("loop", 5, 7), ("4lane", 3, 20), ("minicity", 2, 64)])
def test_done_agents_leave_and_auto_reset(name, E, N, nets, compiled_maps):
    """Episode boundaries (parallel_env.py:303-309; test_parallel_env.py:166-189): with
    max_episode_steps = 5 every agent is done on the 4th action step, the env reports
    dones["__all__"], and the observation handed back is the first one of the next episode."""
    import torch

    from smarts_amd import _native as nat

    sim, ob, cfg = _make(name, E, N, nets, compiled_maps, 31, max_episode_steps=5, auto_reset=True,
                         done_collision=False, done_off_road=False, done_off_route=False)
    first = _host(sim.reset())
    acts = torch.zeros((E, N), dtype=torch.int8, device="cuda")
    for t in range(4):
        out = sim.step(acts)
        torch.cuda.synchronize()
        env_done = out["env_done"].cpu().numpy()
        done = out["done"].cpu().numpy()
        if t < 3:
            assert not env_done.any() and not done.any()
    assert env_done.all() and done.all()
    ev = out["events"].cpu().numpy()
    # the observation is already the reset one: events cleared, trip meter back to zero, all active
    assert (ev[..., nat.EV_REACHED_MAX_EPISODE_STEPS] == 0).all()
    assert out["active"].cpu().numpy().all()
    assert (out["dist"].cpu().numpy() == 0).all()
    assert (sim.env_episode.cpu().numpy() == 1).all()
    # episode 1 starts from spawn row 1
    pos = out["ego_pos"].cpu().numpy().reshape(-1, 3)[:, :2]
    assert np.allclose(pos, sim.spawns[1].cpu().numpy()[:, :2])
    assert not np.allclose(pos, first["ego_pos"][:, :2])
    # the reset observation the tick handed back is what an explicit reset of episode 1 observes:
    # a second batch, reset twice, builds it through the other code path (k_reset + reset pass)
    from smarts_amd.engine import BatchedSim

    twin = BatchedSim(sim.cm, cfg, spawns=sim.spawns.cpu().numpy())
    twin.reset()
    ref = _host(twin.reset())
    got = _host(out)
    for k in ref:
        if k in ("reward", "done", "learner") or k.startswith("final_"):
            continue  # the auto-reset tick keeps the finishing tick's reward / done (and its final rows)
        assert np.array_equal(ref[k], got[k], equal_nan=True), k
    # the finishing tick's low-dimensional rows survived in final_* (smx_outputs): every agent ended on the step limit
    fin = got["final_events"]
    assert (fin[..., nat.EV_REACHED_MAX_EPISODE_STEPS] == 1).all()
    assert (got["final_dist"] > 0).all() and np.isfinite(got["final_ego_pos"]).all()
    assert not np.allclose(got["final_ego_pos"][:, :2], pos)
    twin.close()
    sim.close()


@pytest.mark.parametrize("auto_reset", [False, True])
def test_learner_block_is_reward_and_done(auto_reset, compiled_maps):
    """smx_outputs.learner: float32 [2][E*N] = (reward, done) rewritten whole every tick on
    alternating buffers, zeros for agents that have left."""
    import torch

    from smarts_amd.engine import BatchedSim, SimConfig

    cm = compiled_maps("loop")
    E, N = 4, 8
    sim = BatchedSim(cm, SimConfig(num_envs=E, num_vehicles=N, max_episode_steps=5, auto_reset=auto_reset, num_social=2))
    sim.reset()
    acts = torch.zeros((E, N), dtype=torch.int8, device="cuda")
    seen = set()
    for t in range(12):
        out = sim.step(acts)
        torch.cuda.synchronize()
        block = out["learner"]
        seen.add(block.data_ptr())
        assert torch.equal(block[0], out["reward"].float()), t
        assert torch.equal(block[1], out["done"].float()), t
        assert float(block[:, :, N - 2:].abs().sum()) == 0.0  # social slots
        assert sim.next_learner_block.data_ptr() != block.data_ptr()
    assert len(seen) == 2
    sim.close()


def test_masked_reset_restarts_only_the_selected_envs(compiled_maps):
    """smx_reset with an env mask (a ParallelEnv worker resetting its own env, parallel_env.py:294-296):
    masked envs start their next episode from the next spawn row; the others do not notice."""
    import torch

    from smarts_amd.engine import BatchedSim, SimConfig, make_spawns

    cm = compiled_maps("loop")
    E, N = 4, 6
    cfg = SimConfig(num_envs=E, num_vehicles=N, neighbors=True, nb_radius=50.0)
    spawns = make_spawns(cm, E, N, episodes=3, seed=17)
    sim, twin = BatchedSim(cm, cfg, spawns=spawns), BatchedSim(cm, cfg, spawns=spawns)
    rng = np.random.default_rng(17)
    sim.reset(), twin.reset()
    for t in range(6):
        acts = torch.from_numpy(_actions(rng, E, N)).cuda()
        sim.step(acts), twin.step(acts)
    before = {k: v.clone() for k, v in sim.out.items()}
    mask = torch.tensor([1, 0, 1, 0], dtype=torch.uint8)
    out = sim.reset(mask)
    torch.cuda.synchronize()
    keep = [1, 3]
    for k, v in out.items():
        if k == "learner":
            continue
        assert torch.equal(v[keep], before[k][keep]), f"{k}: an unmasked env changed"
    pos = out["ego_pos"].cpu().numpy().reshape(E, N, 3)
    for e in (0, 2):
        assert np.allclose(pos[e, :, :2], spawns[1, e * N:(e + 1) * N, :2])  # second spawn row
        assert out["active"][e].all() and not out["done"][e].any() and not out["env_done"][e]
        assert int(sim.env_episode[e]) == 1 and int(sim.env_ticks[e]) == cfg.reset_elapsed_steps()
    assert int(sim.env_episode[1]) == 0
    # the unmasked envs keep running exactly like a batch that was never reset
    for t in range(5):
        acts = torch.from_numpy(_actions(rng, E, N)).cuda()
        o1, o2 = sim.step(acts), twin.step(acts)
    torch.cuda.synchronize()
    for k in o1:
        a, b = (o1[k][:, keep], o2[k][:, keep]) if k == "learner" else (o1[k][keep], o2[k][keep])
        assert torch.equal(a, b), k
    sim.close(), twin.close()


def test_full_size_properties(compiled_maps):
    """BASELINE config sizes through size-independent properties: env independence (a shard of
    the batch computes the same thing as the whole batch), determinism, and invariants of the
    dense layout."""
    import torch

    from smarts_amd import _native as nat
    from smarts_amd.engine import BatchedSim, SimConfig, make_spawns

    cm = compiled_maps("loop")
    E, N = 1024, 8
    cfg = SimConfig(num_envs=E, num_vehicles=N, neighbors=True, nb_radius=50.0)
    spawns = make_spawns(cm, E, N, episodes=1, seed=42)
    sim = BatchedSim(cm, cfg, spawns=spawns)
    sub = 64
    cfg2 = SimConfig(num_envs=sub, num_vehicles=N, neighbors=True, nb_radius=50.0)
    sim2 = BatchedSim(cm, cfg2, spawns=spawns[:, (E - sub) * N:])
    sim3 = BatchedSim(cm, cfg, spawns=spawns)
    rng = np.random.default_rng(5)
    sim.reset(), sim2.reset(), sim3.reset()
    for t in range(25):
        acts = torch.from_numpy(_actions(rng, E, N)).cuda()
        o1 = sim.step(acts)
        o2 = sim2.step(acts[E - sub:].contiguous())
        o3 = sim3.step(acts)
    torch.cuda.synchronize()
    for k in o1:
        a, b, c = o1[k].cpu().numpy(), o2[k].cpu().numpy(), o3[k].cpu().numpy()
        assert np.array_equal(a, c, equal_nan=True), f"{k}: two identical runs differ"
        part = a[:, E - sub:] if k == "learner" else a[E - sub:]  # the learner block is [2, E, N]
        assert np.array_equal(part, b, equal_nan=True), f"{k}: env results depend on batch placement"
    act = o1["active"].cpu().numpy().astype(bool)
    wpc = o1["wp_count"].cpu().numpy()
    assert (wpc[act][:, 0] >= 1).all() and (wpc[act][:, 1] == cfg.wp_len).all()
    hd = o1["wp_heading"].cpu().numpy()
    assert (np.abs(hd) <= np.pi + 1e-6).all()
    lane = o1["ego_lane"].cpu().numpy()
    assert (lane[act][:, 0] >= 0).all() and (lane[act][:, 0] < cm.n_lanes).all()
    nbc = o1["nb_count"].cpu().numpy()
    assert (nbc <= N - 1).all()
    slots = o1["nb_slot"].cpu().numpy()
    assert ((slots >= -1) & (slots < N)).all()
    # rewards are trip-meter increments (metres along lane 0 of the current road per tick): at
    # 15 m/s and dt = 0.1 the typical value is ~1.5; the reference's own bound for a single Laner
    # agent is (-3, 3) (test_hiway_env.py:53-61); hops between roads may add a few metres
    r = o1["reward"].cpu().numpy()
    assert (np.abs(r) < 25).all()
    assert 0.5 < np.median(r[act]) < 2.5
    assert np.isfinite(o1["ego_pos"].cpu().numpy()).all()
    for s in (sim, sim2, sim3):
        s.close()


def test_large_batch_launch_strategy_agrees_with_small_batch(compiled_maps):
    """Above 16384 vehicles (the LARGE launch form) the scan halves, the OGM tiles and every sensor role get their own
    launch (smx_kernels.hip enqueue()); a 4160 x 8 batch must compute what a 32-env slice of it
    (small-batch strategy) computes."""
    import torch

    from smarts_amd.engine import BatchedSim, SimConfig, make_spawns

    cm = compiled_maps("4lane")
    E, N, sub = 4160, 8, 32
    spawns = make_spawns(cm, sub, N, episodes=1, seed=9)
    big = np.tile(spawns, (1, E // sub, 1))  # the slice repeated: every env group sees the same worlds
    kw = dict(num_vehicles=N, neighbors=True, nb_radius=50.0, ogm=True, ogm_width=64, ogm_height=64, ogm_resolution=50 / 64)
    sim = BatchedSim(cm, SimConfig(num_envs=E, **kw), spawns=big)
    sim2 = BatchedSim(cm, SimConfig(num_envs=sub, **kw), spawns=spawns)
    rng = np.random.default_rng(9)
    sim.reset(), sim2.reset()
    for t in range(12):
        a_small = _actions(rng, sub, N)
        o1 = sim.step(torch.from_numpy(np.tile(a_small, (E // sub, 1))).cuda())
        o2 = sim2.step(torch.from_numpy(a_small).cuda())
    torch.cuda.synchronize()
    for k in o2:
        a, b = o1[k].cpu().numpy(), o2[k].cpu().numpy()
        first, last = (a[:, :sub], a[:, E - sub:]) if k == "learner" else (a[:sub], a[E - sub:])
        assert np.array_equal(first, b, equal_nan=True) and np.array_equal(last, b, equal_nan=True), k
    sim.close(), sim2.close()


def test_step_before_reset_raises(compiled_maps):
    import torch

    from smarts_amd.engine import BatchedSim, SimConfig

    sim = BatchedSim(compiled_maps("loop"), SimConfig(num_envs=1, num_vehicles=2))
    with pytest.raises(RuntimeError):
        sim.step(torch.zeros((1, 2), dtype=torch.int8, device="cuda"))
    sim.close()
