"""GPU parity of the LARGE launch strategy (smx_set_launch_strategy): one launch per role, k_scan's halves back
to back, the register form of k_control, waypoint rows emitted in memory order from LDS knot tables
(k_waypoints_tables), k_lidar / k_ogm on their own.  AUTO picks this form above 16384 vehicles, where an oracle
run is out of reach; forced onto oracle-sized batches here, every output is held to the oracle directly, and
BASELINE-shaped batches above the threshold are held to small-strategy slices of themselves.
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
    return np.where(rng.random((E, N)) < 0.8, 0, rng.integers(1, 4, (E, N))).astype(np.int8)


def _make(name, E, N, nets, compiled_maps, seed, **cfg_kw):
    from smarts_amd.engine import BatchedSim, SimConfig, make_spawns

    cm = compiled_maps(name)
    cfg = SimConfig(num_envs=E, num_vehicles=N, neighbors=True, nb_radius=50.0, **cfg_kw)
    spawns = make_spawns(cm, E, N, episodes=2, seed=seed)
    sim = BatchedSim(cm, cfg, spawns=spawns)
    ob = parity.OracleBatch(nets(name), cm, cfg, spawns[0])
    return sim, ob, cfg


@pytest.mark.parametrize("name,E,N,T,seed,extra", [
    ("loop", 5, 8, 50, 111, {}),                      # 5 envs: the last workgroup of every role is ragged
    ("4lane", 3, 16, 50, 112, {}),                    # junction branchings inside the lookahead: the long-way numbering
    ("minicity", 2, 16, 35, 113, {}),                 # long knot runs: rows that leave through the serial emitter
    ("loop", 2, 32, 25, 114, dict(ogm=True, ogm_width=64, ogm_height=64, ogm_resolution=50 / 64)),  # C4's shape
    ("minicity", 1, 64, 12, 115, dict(lidar="planar100")),  # C5's shape: 64-vehicle envs, k_lidar on its own
    ("loop", 3, 8, 30, 116, dict(wp_paths=2, wp_len=33)),   # rows of the whole lookahead, fewer rows than lanes
    ("4lane", 2, 8, 20, 117, dict(wp_paths=8, wp_len=10)),  # more rows than a road has lanes
])
@pytest.mark.parametrize("cut", ["large", "large_one_lane"])
def test_large_strategy_teacher_forced_against_the_oracle(name, E, N, T, seed, extra, cut, nets, compiled_maps):
    """Both cuts of the LARGE form on every map: "large" picks by the map (teams where lanes split, else one lane per
    vehicle + slow lists — with the team kernel for the path seeds at these sizes), "large_one_lane" forces the one-lane
    cut with the one-lane seeds kernel and its slow chain — on 4lane / minicity a third of the vehicles then goes
    through the slow lists (branchings, junction roads, new roads)."""
    import torch

    from smarts_amd.lidar import Planar100

    extra = dict(extra)
    if extra.get("lidar") == "planar100":
        extra["lidar"] = Planar100
    sim, ob, cfg = _make(name, E, N, nets, compiled_maps, seed, launch_strategy=cut, **extra)
    assert sim.launch_form() == ("large_one_lane" if (cut == "large_one_lane" or name == "loop") else "large_teams")
    d, o = _host(sim.reset()), ob.reset_observe()
    assert parity.compare(d, o, tol64=1e-9, tol32=2e-6, where="reset ") == []
    rng = np.random.default_rng(seed)
    for t in range(T):
        acts = _actions(rng, E, N)
        if t % 5 == 2:
            acts[0, 0] = -1
        d, o = _host(sim.step(torch.from_numpy(acts).cuda())), ob.step(acts)
        bad = parity.compare(d, o, tol64=1e-9, tol32=2e-5, where=f"{name} large t{t} ")
        assert bad == [], "\n".join(bad[:8])
        parity.sync_oracle_from_device(ob, sim)
    sim.close()


@pytest.mark.parametrize("name,N,extra", [
    ("loop", 8, {}), ("4lane", 16, {}), ("minicity", 16, {}),
    ("minicity", 64, dict(lidar="planar100")),
    ("loop", 32, dict(ogm=True, ogm_width=64, ogm_height=64, ogm_resolution=50 / 64)),
])
def test_strategies_agree_bit_for_bit(name, N, extra, compiled_maps):
    """SMALL and both cuts of LARGE run the same arithmetic over different launches: 40 auto-reset ticks of the same
    batch must leave identical bits in every output and in the whole state."""
    import torch

    from smarts_amd.engine import BatchedSim, SimConfig, make_spawns
    from smarts_amd.lidar import Planar100

    extra = dict(extra)
    if extra.get("lidar") == "planar100":
        extra["lidar"] = Planar100
    cm = compiled_maps(name)
    E = max(2, 192 // N)
    spawns = make_spawns(cm, E, N, episodes=3, seed=21)
    sims = [BatchedSim(cm, SimConfig(num_envs=E, num_vehicles=N, neighbors=True, nb_radius=50.0, auto_reset=True,
                                     launch_strategy=s, **extra), spawns=spawns) for s in ("small", "large", "large_one_lane", "large_teams")]
    assert [s.launch_form() for s in sims] == ["small", "large_one_lane" if name == "loop" else "large_teams", "large_one_lane", "large_teams"]
    rng = np.random.default_rng(21)
    for s in sims:
        s.reset()
    for t in range(40):
        acts = torch.from_numpy(_actions(rng, E, N)).cuda()
        outs = [s.step(acts) for s in sims]
        if t % 13 == 0 or t == 39:
            torch.cuda.synchronize()
            for other in (1, 2, 3):
                for k in outs[0]:
                    assert np.array_equal(outs[0][k].cpu().numpy(), outs[other][k].cpu().numpy(), equal_nan=True), (t, k, other)
                assert np.array_equal(sims[0].state.cpu().numpy(), sims[other].state.cpu().numpy(), equal_nan=True), (t, other)
                assert np.array_equal(sims[0].flags.cpu().numpy(), sims[other].flags.cpu().numpy()), (t, other)
    for s in sims:
        s.close()


@pytest.mark.parametrize("name,E,N,sub,ticks,extra", [
    # BASELINE configs[3] shape: loop, 32-vehicle envs + OGM 64 x 64, far above the 16384-vehicle threshold
    ("loop", 1056, 32, 4, 10, dict(ogm=True, ogm_width=64, ogm_height=64, ogm_resolution=50 / 64)),
    # BASELINE configs[4] shape: minicity, 64-vehicle envs + 100-ray lidar (k_lidar on its own, k_scan<false> on the big map)
    ("minicity", 520, 64, 2, 8, dict(lidar="planar100")),
    # BASELINE configs[2] shape: 4lane, 16-vehicle envs, collisions
    ("4lane", 2056, 16, 8, 10, {}),
    # just above the threshold (16 640 vehicles), and between the LDS-path limit of k_control (8 192) and the
    # threshold: the small form with the register form of k_control
    ("loop", 520, 32, 4, 10, dict(ogm=True, ogm_width=64, ogm_height=64, ogm_resolution=50 / 64)),
    ("4lane", 768, 16, 8, 10, {}),
])
def test_batches_above_the_threshold_agree_with_small_slices(name, E, N, sub, ticks, extra, compiled_maps):
    """AUTO above 16384 vehicles = the LARGE form (below: the small form, past 8192 vehicles with the register
    form of k_control).  The batch is `sub` distinct envs tiled E / sub times, so the
    first and the last slice must both equal a `sub`-env batch stepped in the small form."""
    import torch

    from smarts_amd.engine import BatchedSim, SimConfig, make_spawns
    from smarts_amd.lidar import Planar100

    extra = dict(extra)
    if extra.get("lidar") == "planar100":
        extra["lidar"] = Planar100
    assert E * N > 8192 and E % sub == 0
    cm = compiled_maps(name)
    spawns = make_spawns(cm, sub, N, episodes=1, seed=9)
    big = np.tile(spawns, (1, E // sub, 1))
    kw = dict(num_vehicles=N, neighbors=True, nb_radius=50.0, **extra)
    sim = BatchedSim(cm, SimConfig(num_envs=E, **kw), spawns=big)
    sim2 = BatchedSim(cm, SimConfig(num_envs=sub, launch_strategy="small", **kw), spawns=spawns)
    rng = np.random.default_rng(9)
    sim.reset(), sim2.reset()
    for t in range(ticks):
        a_small = _actions(rng, sub, N)
        o1 = sim.step(torch.from_numpy(np.tile(a_small, (E // sub, 1))).cuda())
        o2 = sim2.step(torch.from_numpy(a_small).cuda())
    torch.cuda.synchronize()
    for k in o2:
        a, b = o1[k].cpu().numpy(), o2[k].cpu().numpy()
        first, last = (a[:, :sub], a[:, E - sub:]) if k == "learner" else (a[:sub], a[E - sub:])
        assert np.array_equal(first, b, equal_nan=True) and np.array_equal(last, b, equal_nan=True), k
    sim.close(), sim2.close()


def test_the_large_form_picks_its_cut_by_map_and_size(compiled_maps):
    """smx_launch_form: one lane per vehicle + slow lists where the map's lanes never split (loop, whose junction-internal
    connector lanes have one successor each); teams on maps with branchings."""
    from smarts_amd.engine import BatchedSim, SimConfig, make_spawns

    def form(name, E, N, strategy="auto"):
        cm = compiled_maps(name)
        sim = BatchedSim(cm, SimConfig(num_envs=E, num_vehicles=N, launch_strategy=strategy),
                         spawns=make_spawns(cm, E, N, episodes=1, seed=5))
        f = sim.launch_form()
        sim.close()
        return f

    assert form("loop", 64, 8) == "small"
    assert form("loop", 1024, 32) == "large_one_lane"
    assert form("4lane", 4200, 16) == "large_teams"
    assert form("4lane", 4200, 16, "large_one_lane") == "large_one_lane"
    assert form("loop", 2048, 32, "large_teams") == "large_teams"


def test_rows_from_the_overflow_area_equal_the_small_forms(compiled_maps):
    """k_waypoints_emit keeps the knot records of its workgroup's paths in an LDS pool; the paths that do not fit go
    through an overflow area in device memory and a sweep of their own.  With the pool cut to 40 records nearly every
    workgroup overflows: every row must still equal the SMALL form's, bit for bit, over auto-reset ticks."""
    import ctypes as C

    import torch

    from smarts_amd.engine import BatchedSim, SimConfig, make_spawns

    for name, N in (("loop", 32), ("minicity", 16)):
        cm = compiled_maps(name)
        E = 256 // N
        spawns = make_spawns(cm, E, N, episodes=3, seed=31)
        sims = [BatchedSim(cm, SimConfig(num_envs=E, num_vehicles=N, neighbors=True, nb_radius=50.0, auto_reset=True,
                                         launch_strategy=s), spawns=spawns) for s in ("small", "large_one_lane")]
        sims[1].lib.smx_debug_set_wp_pool.argtypes = [C.c_void_p, C.c_int32]
        assert sims[1].lib.smx_debug_set_wp_pool(sims[1].handle, 40) == 0
        rng = np.random.default_rng(31)
        for s in sims:
            s.reset()
        for t in range(60):
            acts = torch.from_numpy(_actions(rng, E, N)).cuda()
            outs = [s.step(acts) for s in sims]
            if t % 7 == 0 or t == 59:
                torch.cuda.synchronize()
                for k in outs[0]:
                    assert np.array_equal(outs[0][k].cpu().numpy(), outs[1][k].cpu().numpy(), equal_nan=True), (name, t, k)
                assert np.array_equal(sims[0].state.cpu().numpy(), sims[1].state.cpu().numpy(), equal_nan=True), (name, t)
        for s in sims:
            s.close()
