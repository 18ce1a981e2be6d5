"""Multi-GPU: one process per GPU, each owning a contiguous range of environment instances.

Environment instances are closed worlds (the reference runs each in its own process with its
own seed ``seed + i``, ``smarts/env/wrappers/parallel_env.py:96-122,190-202``), so the data
path has **no collective**: every rank steps its own shard.  The only exchange is the optional
learner-side gather of the small per-tick ``{reward, done}`` block, done with
``torch.distributed`` (backend ``"nccl"`` = RCCL over xGMI on ROCm; ``"gloo"`` on CPU in tests).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class ShardPlan:
    """Env range ``[first_env, first_env + num_envs)`` of ``rank`` (SURVEY.md §8e partitioning)."""

    total_envs: int
    world_size: int
    rank: int

    @property
    def first_env(self) -> int:
        base, rem = divmod(self.total_envs, self.world_size)
        return self.rank * base + min(self.rank, rem)

    @property
    def num_envs(self) -> int:
        base, rem = divmod(self.total_envs, self.world_size)
        return base + (1 if self.rank < rem else 0)

    def seed_of(self, seed: int, local_env: int) -> int:
        """ParallelEnv.seed: env i gets ``seed + i`` over the *global* index (parallel_env.py:199)."""
        return seed + self.first_env + local_env


def rank_environments(world: int, port: Optional[int] = None, base=None):
    """The environment of every rank of a one-node job, as ``torch.distributed.run`` would set it (one process per
    GPU): RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR = 127.0.0.1, MASTER_PORT (a free port when none is given)."""
    import socket

    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    base = dict(os.environ if base is None else base)
    envs = []
    for r in range(world):
        e = dict(base)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        envs.append(e)
    return envs


def self_launch(argv, world: int, port: Optional[int] = None) -> int:
    """Start ``world`` ranks of the command ``argv`` (one child process each, the reference's own data parallelism:
    parallel_env.py:96-122) and wait for them; the caller has not touched the GPU (a process that has must not be
    replaced or forked).  Rank 0 inherits stdout — its one JSON line is the job's; the other ranks' stdout is dropped.
    Returns the largest exit code."""
    import subprocess

    procs = []
    for r, env in enumerate(rank_environments(world, port)):
        procs.append(subprocess.Popen(list(argv), env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    codes = [p.wait() for p in procs]
    return max((c if c >= 0 else 128 - c) for c in codes)


def env_from_dist() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process if absent)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_process_group(backend: Optional[str] = None):
    rank, local_rank, world = env_from_dist()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("SMX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            # one process per GPU: bind the rank to its device before the first collective
            torch.cuda.set_device(local_device_index(local_rank))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def local_device_index(local_rank: int) -> int:
    """The GPU of this rank: ``local_rank`` (one process per GPU).  ``SMX_REHEARSE_ONE_GPU=1`` folds
    the ranks onto the devices that exist — a rehearsal of the multi-rank path on a one-GPU box
    (with ``SMX_DIST_BACKEND=gloo``; RCCL refuses two ranks on one device), never a measurement."""
    if os.environ.get("SMX_REHEARSE_ONE_GPU") == "1":
        return local_rank % max(torch.cuda.device_count(), 1)
    return local_rank


def barrier():
    """dist.barrier that names the rank's device under NCCL (avoids the device guess)."""
    if not dist.is_initialized():
        return
    if dist.get_backend() == "nccl":
        dist.barrier(device_ids=[torch.cuda.current_device()])
    else:
        dist.barrier()


class RewardDoneGather:
    """Per-tick gather of the learner-facing block: reward (f32) and done (u8) of every agent of
    every shard, packed as one f32 tensor ``[2, E_shard * N]`` so that it is ONE small collective
    (C4: 4096 x 32 agents x 8 B = 1 MB node-wide).  Requires equal shard sizes (all_gather).

    The collective is off the simulation's critical path: ``start`` packs the block on the current
    stream and launches the all-gather asynchronously (RCCL runs it on its own stream, ordered after
    the packing), so the next tick's kernels do not wait for it; two buffer pairs alternate, and a
    pair is waited for only before it is reused two ticks later or when ``result`` is asked for.
    ``__call__`` = ``start`` + ``result`` (synchronous form)."""

    def __init__(self, num_envs: int, num_vehicles: int, device, world_size: int):
        self.world = world_size
        self.n = num_envs * num_vehicles
        self.send = [torch.zeros((2, self.n), dtype=torch.float32, device=device) for _ in range(2)]
        self.recv = [torch.zeros((world_size, 2, self.n), dtype=torch.float32, device=device) for _ in range(2)]
        self.work = [None, None]
        self.src = [None, None]  # data_ptr of the caller's block a pending gather reads from
        self.k = 0  # pair used by the latest start()

    def _wait(self, k: int):
        if self.work[k] is not None:
            self.work[k].wait()  # NCCL: makes the current stream wait; gloo: blocks the host
            self.work[k] = None
        self.src[k] = None

    def release(self, block: torch.Tensor) -> None:
        """Call before enqueuing work that overwrites ``block``: waits for a gather still reading it."""
        for k in (0, 1):
            if self.src[k] is not None and self.src[k] == block.data_ptr():
                self._wait(k)

    def start(self, reward: torch.Tensor, done: torch.Tensor) -> None:
        k = self.k ^ 1
        self._wait(k)  # the gather launched two ticks ago on this pair
        self.send[k][0].copy_(reward.reshape(-1))
        self.send[k][1].copy_(done.reshape(-1))
        if self.world > 1:
            self.work[k] = dist.all_gather_into_tensor(self.recv[k].view(-1), self.send[k].view(-1), async_op=True)
        else:
            self.recv[k][0].copy_(self.send[k])
        self.k = k

    def start_packed(self, block: torch.Tensor) -> None:
        """The same for a block the kernels already packed (``BatchedSim.out["learner"]``,
        float32 ``[2, E, N]``): no packing kernels; the block must stay untouched until the gather
        launched from it has been waited for (BatchedSim alternates two of them)."""
        k = self.k ^ 1
        self._wait(k)
        flat = block.reshape(-1)
        if self.world > 1:
            self.work[k] = dist.all_gather_into_tensor(self.recv[k].view(-1), flat, async_op=True)
            self.src[k] = block.data_ptr()
        else:
            self.recv[k] = block.reshape(1, 2, self.n)  # single process: the block itself
        self.k = k

    def result(self) -> torch.Tensor:
        """``[world, 2, E_shard * N]`` of the latest ``start`` (waits for it)."""
        self._wait(self.k)
        return self.recv[self.k]

    def finish(self) -> None:
        self._wait(0)
        self._wait(1)

    def __call__(self, reward: torch.Tensor, done: torch.Tensor) -> torch.Tensor:
        self.start(reward, done)
        return self.result()
