"""Factories with the reference's names (gym_os2r/common/__init__.py:12-53).  ``make_mp_envs``
returns one batched VecEnv on one GPU instead of ``nenvs`` OS processes."""
import functools

from ..registry import make
from .vec_env import HipVecEnv


def make_env_from_id(env_id: str, **kwargs):
    return make(env_id, **kwargs)


def make_mp_envs(env_id, nenvs, seed, randomizer, start_idx=0, num_splits=1, **kwargs):
    """``nenvs`` environments with per-env RNG streams keyed by ``seed`` and the global env index
    ``start_idx + i`` (the reference seeds process ``i`` with ``seed + start_idx + i``).
    ``num_splits`` > 1 cuts the batch into that many contiguous shards with a handle and a stream each
    (common/vec_env.py); the results are those of the single batch, bit for bit."""
    from ..distributed import shard_range
    if int(num_splits) > 1:
        # one hardware queue per shard stream: the HIP runtime maps streams onto GPU_MAX_HW_QUEUES queues (4 by default)
        # and shards that share a queue run one after the other; the variable is read when the runtime initialises
        import os
        import warnings
        import torch
        if "GPU_MAX_HW_QUEUES" not in os.environ:
            if torch.cuda.is_initialized():
                warnings.warn("num_splits > 1 after the GPU was initialised: set GPU_MAX_HW_QUEUES >= 2 * num_splits in the "
                              "environment so that every shard stream gets a hardware queue of its own")
            else:
                os.environ["GPU_MAX_HW_QUEUES"] = str(max(8, 2 * int(num_splits)))
    shards = []
    for r in range(int(num_splits)):
        offset, count = shard_range(nenvs, r, int(num_splits))
        make_env = functools.partial(make_env_from_id, env_id=env_id, num_envs=count, seed=seed,
                                     env_offset=start_idx + offset, **kwargs)
        shards.append(randomizer(env=make_env))
    return HipVecEnv(*shards)
