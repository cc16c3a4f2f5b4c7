"""World-size-2 checks on CPU (gloo): shard arithmetic, global-order gather, and that a sharded
job reproduces the single-handle result env for env (per-env RNG keyed by the global index)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from gym_os2r_amd.distributed import shard_range


def test_shard_range_partitions():
    for total in (1, 7, 64, 65536, 524288, 1001):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (o1, c1), (o2, _) in zip(spans, spans[1:]):
                assert o1 + c1 == o2
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    assert shard_range(524288, 3, 8) == (3 * 65536, 65536)
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import make_config
    from gym_os2r_amd import abi
    from gym_os2r_amd.distributed import gather_to_rank0, rank_world, shard_range
    from oracle import oracle_py
    assert rank_world() == (rank, world, rank)
    off, cnt = shard_range(total, rank, world)
    # the oracle stands in for the device here: same env_offset / seed semantics as the C-ABI
    cfg, _, _ = make_config("fixed_hip", "BalancingV2", True, num_envs=cnt, env_offset=off, seed=5,
                            reset_mode=abi.RESET_RANDOM, randomize_params=True, max_episode_steps=4)
    sim = oracle_py.OracleSim(cfg)
    for _ in range(8):
        obs, rew, done, _ = sim.step(None)           # counter-RNG actions keyed by the global env index
    dist.barrier()                                   # no collective on the step path; this is the bench bracket
    g_obs = gather_to_rank0(torch.from_numpy(obs), total, key="obs")
    g_done = gather_to_rank0(torch.from_numpy(done), total, key="done")
    # the gather writes into one preallocated output per key: a second call returns the same storage, refilled
    again = gather_to_rank0(torch.from_numpy(obs) * 0 + float(rank), total, key="obs")
    if rank == 0:
        assert again.data_ptr() == g_obs.data_ptr() and float(again[0, 0]) == 0.0 and float(again[-1, 0]) == float(world - 1)
    g_obs = gather_to_rank0(torch.from_numpy(obs), total, key="obs")
    # without a key every call returns its own tensor: two anonymous call sites cannot alias each other
    anon1 = gather_to_rank0(torch.from_numpy(obs), total)
    anon2 = gather_to_rank0(torch.from_numpy(obs) * 0, total)
    if rank == 0:
        assert anon1.data_ptr() != anon2.data_ptr() and torch.equal(anon1, g_obs) and float(anon2.abs().max()) == 0.0
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)         # max-over-ranks timing reduction used by bench.py
    assert float(t) == world
    if rank == 0:
        q.put((g_obs.numpy(), g_done.numpy()))
    else:
        assert g_obs is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total,world,port", [(37, 2, 29533), (36, 2, 29535), (9, 1, 29537), (61, 8, 29539), (64, 8, 29551)])
def test_two_rank_shards_reproduce_the_single_handle_run(oracle, total, world, port):
    """uneven shards (19 + 18: padded staging block), even shards (the collective writes straight into the output's slices),
    a single rank (the collective still runs: what the one-GPU rehearsal of bench.py executes), and WORLD SIZE 8 -- the size of
    the first 8-GPU run -- with uneven (5 x 8 + 3 x 7) and even shards"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    g_obs, g_done = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    from helpers import make_config
    from gym_os2r_amd import abi
    cfg, _, _ = make_config("fixed_hip", "BalancingV2", True, num_envs=total, env_offset=0, seed=5,
                            reset_mode=abi.RESET_RANDOM, randomize_params=True, max_episode_steps=4)
    sim = oracle.OracleSim(cfg)
    for _ in range(8):
        obs, rew, done, _ = sim.step(None)
    assert np.array_equal(g_obs, obs) and np.array_equal(g_done, done)
    assert (done != 0).any()                          # the TimeLimit fired: resets were exercised too
