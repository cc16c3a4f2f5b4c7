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


class Rank0Gather:
    """Gathers per-rank [n_r, ...] tensors into ONE preallocated [total_envs, ...] tensor on rank ``dst``, in global env
    order -- what replaces the ``np.stack`` of the workers' results on the parent process in the reference
    (common/vec_env/subproc_vec_env.py:119-123).

    Nothing is allocated or concatenated per call: rank ``dst`` owns the output (one per ``key``: obs / reward / done ...,
    or the caller's own ``out=``) and the collective -- ``torch.distributed.gather``: RCCL sends and receives over xGMI for
    CUDA tensors ("nccl" backend), gloo for CPU tensors -- writes every rank's rows straight into their slice of it.
    Shards that differ by one row (``shard_range``) go through a preallocated padded staging block, because the collective
    wants equal sizes, and are copied slice by slice from there.  Works with a single rank too (the collective runs; the
    one-GPU rehearsal of the multi-rank path depends on that).  Stream-ordered on the current stream, as the collective is.
    """

    def __init__(self, total_envs: int, dst: int = 0):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("Rank0Gather needs an initialised process group")
        self.dst, self.total = int(dst), int(total_envs)
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        parts = [shard_range(self.total, r, self.world) for r in range(self.world)]
        self.offsets, self.counts = [p[0] for p in parts], [p[1] for p in parts]
        self.nmax = max(self.counts)
        self.even = min(self.counts) == self.nmax
        self._out, self._stage, self._pad = {}, {}, {}

    def _like(self, tensor, rows):
        import torch
        return torch.empty((rows,) + tuple(tensor.shape[1:]), dtype=tensor.dtype, device=tensor.device)

    def __call__(self, tensor, key=None, out=None):
        """-> the gathered [total_envs, ...] tensor on rank ``dst``, None elsewhere.  With a ``key`` the output is the gatherer's own
        buffer for that key (the same storage every call: two call sites must not share a key); without one, and without ``out``,
        a fresh tensor per call."""
        import torch.distributed as dist
        if tensor.shape[0] != self.counts[self.rank]:
            raise ValueError(f"rank {self.rank} owns {self.counts[self.rank]} environments, got {tensor.shape[0]} rows")
        tensor = tensor.contiguous()
        sig = (key, tuple(tensor.shape[1:]), tensor.dtype, tensor.device)
        on_dst = self.rank == self.dst
        if on_dst and out is None:
            out = self._out.get(sig) if key is not None else None
            if out is None:
                out = self._like(tensor, self.total)
                if key is not None:
                    self._out[sig] = out
        if self.even:
            views = [out[o:o + c] for o, c in zip(self.offsets, self.counts)] if on_dst else None
            dist.gather(tensor, views, dst=self.dst)
            return out if on_dst else None
        send = tensor
        if tensor.shape[0] < self.nmax:                # (one row short: padded in a block that is allocated once)
            send = self._pad.get(sig)
            if send is None:
                send = self._pad[sig] = self._like(tensor, self.nmax).zero_()
            send[:tensor.shape[0]].copy_(tensor)
        stage = None
        if on_dst:
            stage = self._stage.get(sig)
            if stage is None:
                stage = self._stage[sig] = self._like(tensor, self.world * self.nmax)
        dist.gather(send, [stage[r * self.nmax:(r + 1) * self.nmax] for r in range(self.world)] if on_dst else None, dst=self.dst)
        if not on_dst:
            return None
        for r, (o, c) in enumerate(zip(self.offsets, self.counts)):
            out[o:o + c].copy_(stage[r * self.nmax:r * self.nmax + c])
        return out


_gatherers = {}


def gather_to_rank0(tensor, total_envs: int, dst: int = 0, key=None, out=None):
    """Per-rank [n_r, ...] tensors -> one [total_envs, ...] tensor on rank ``dst`` in global env order (None elsewhere):
    ``Rank0Gather`` with one cached instance per (process group, total_envs, dst) -- a group that is destroyed and initialised
    again, with another rank or backend, gets a new one.  With a ``key`` the result on rank ``dst`` is the gatherer's own buffer
    for that key -- overwritten by the next gather with the same key, shape and dtype --; without one (and without ``out``) it is
    a fresh tensor.  Without a process group the tensor is returned as it is."""
    import torch.distributed as dist
    if not dist.is_initialized():
        return tensor
    pg = dist.distributed_c10d._get_default_group()
    ident = (id(pg), dist.get_backend(), dist.get_rank(), dist.get_world_size(), int(total_envs), int(dst))
    g = _gatherers.get(ident)
    if g is None:
        for k in [k for k in _gatherers if k[0] != id(pg)]:      # buffers of groups that are gone
            del _gatherers[k]
        g = _gatherers[ident] = Rank0Gather(total_envs, dst)
        g._pg = pg                                     # (keeps the group object alive, so its id cannot be reused while cached)
    return g(tensor, key=key, out=out)
