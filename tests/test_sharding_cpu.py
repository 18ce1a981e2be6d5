"""N > 1 path on CPU: shard plan + the reward/done gather over gloo with world_size 2."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from smarts_amd.sharding import RewardDoneGather, ShardPlan


def test_shard_plan_partitions_every_env_once():
    for total, world in [(4096, 8), (1024, 1), (10, 4), (7, 8)]:
        seen = []
        for r in range(world):
            p = ShardPlan(total, world, r)
            seen += list(range(p.first_env, p.first_env + p.num_envs))
        assert seen == list(range(total))
    # per-env seeds follow the global index like ParallelEnv.seed (parallel_env.py:190-202;
    # test_parallel_env.py:104-112: seeds = first + i)
    p = ShardPlan(4096, 8, 3)
    assert p.first_env == 1536 and p.num_envs == 512
    assert p.seed_of(42, 0) == 42 + 1536 and p.seed_of(42, 5) == 42 + 1541


def test_spawns_are_shard_invariant(compiled_maps):
    """A rank's spawn table equals the slice of the single-process table: sharding changes where an
    env runs, not what it computes."""
    from smarts_amd.engine import make_spawns

    cm = compiled_maps("loop")
    whole = make_spawns(cm, 8, 4, episodes=2, seed=42)
    for r in range(2):
        p = ShardPlan(8, 2, r)
        part = make_spawns(cm, p.num_envs, 4, episodes=2, seed=42, first_env=p.first_env)
        assert np.array_equal(part, whole[:, p.first_env * 4:(p.first_env + p.num_envs) * 4])


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    E, N = 3, 4
    g = RewardDoneGather(E, N, "cpu", world)
    reward = torch.arange(E * N, dtype=torch.float64).reshape(E, N) + 100 * rank
    done = ((torch.arange(E * N) + rank) % 2).to(torch.uint8).reshape(E, N)
    out = g(reward, done).clone()
    # pipelined form: three ticks in flight over the two buffer pairs, results read one tick late
    seen = []
    for t in range(3):
        g.start(reward + t, done)
        seen.append(g.result()[:, 0, 0].clone())
    g.finish()
    assert all(float(seen[t][src]) == t + 100 * src for t in range(3) for src in range(world))
    # packed form: the caller's own alternating blocks are gathered without a packing copy
    blocks = [torch.zeros((2, E, N), dtype=torch.float32), torch.zeros((2, E, N), dtype=torch.float32)]
    for t in range(4):
        b = blocks[t % 2]
        g.release(b)
        b[0].fill_(10 * t + rank)
        b[1].fill_(rank)
        g.start_packed(b)
        got = g.result()
        assert all(float(got[src, 0, 0]) == 10 * t + src and float(got[src, 1, -1]) == src for src in range(world))
    g.finish()
    q.put((rank, out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_reward_done_gather_gloo_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        got = results[r]
        assert got.shape == (2, 2, 12)
        for src in range(world):
            assert np.array_equal(got[src, 0], np.arange(12) + 100 * src)
            assert np.array_equal(got[src, 1], (np.arange(12) + src) % 2)


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the shape of the driver's one-GPU command)
    must start two ranks itself — the reference's data parallelism is one process per env group
    (smarts/env/wrappers/parallel_env.py:96-122) — and rank 0 must print the job's line.  --launch-check keeps the
    children off the GPU: rendezvous over gloo, all-reduce of the ranks, {n_gpus, rank_sum}."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    for n in (2, 3):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--launch-check"], env=env,
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, out.stdout  # one line for the job, from rank 0
        rec = json.loads(lines[0])
        assert rec["n_gpus"] == n and rec["rank_sum"] == n * (n - 1) / 2


def test_rank_environments_are_torchrun_shaped():
    from smarts_amd.sharding import rank_environments

    envs = rank_environments(4, port=29999, base={})
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29999" for e in envs)
