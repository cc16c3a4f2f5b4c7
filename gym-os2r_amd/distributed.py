"""Multi-GPU sharding: one process per GPU, contiguous env ranges, no collective on the step path.

The reference scales by running one OS process per environment (common/vec_env/
subproc_vec_env.py:98-108, seeds ``seed + rank``); environments are independent, so here rank r
of W owns the global env indices [r*n, (r+1)*n) and keys its counter RNG with the global index
(``env_offset``): a W-rank job produces bit-identical per-env trajectories to a single handle
with W*n environments.  The only communication offered is an optional observation gather to
rank 0 (RCCL over xGMI under the "nccl" backend); it is off the step path.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple


def rank_world() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torch.distributed launcher environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def shard_range(total_envs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced partition: (offset, count) of this rank."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    base, rem = divmod(int(total_envs), int(world))
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def make_sharded(env_id: str, total_envs: int, seed: int = 0, randomizer=None, **kwargs):
    """This rank's shard of a ``total_envs``-environment job as one batched runtime."""
    from .registry import make
    rank, world, local_rank = rank_world()
    offset, count = shard_range(total_envs, rank, world)
    kwargs.setdefault("device", f"cuda:{local_rank}")
    factory = lambda: make(env_id, num_envs=count, seed=seed, env_offset=offset, **kwargs)  # noqa: E731
    return randomizer(env=factory) if randomizer is not None else factory()


def gather_to_rank0(tensor, total_envs: int, dst: int = 0):
    """Concatenate per-rank [n_r, ...] tensors on rank ``dst`` in global env order (None elsewhere).

    Uses ``torch.distributed.gather`` on the initialised process group: RCCL over xGMI for CUDA
    tensors ("nccl" backend), gloo for CPU tensors.  Shards may differ by one row; they are padded
    to the largest shard for the collective and trimmed afterwards.
    """
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return tensor
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [shard_range(total_envs, r, world)[1] for r in range(world)]
    nmax = max(counts)
    pad = tensor
    if tensor.shape[0] < nmax:
        pad = torch.cat([tensor, tensor.new_zeros((nmax - tensor.shape[0],) + tuple(tensor.shape[1:]))])
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad.contiguous(), bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)])
