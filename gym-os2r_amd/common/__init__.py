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
    shards = []
    for r in range(int(num_splits)):
        offset, count = shard_range(nenvs, r, int(num_splits))
        make_env = functools.partial(make_env_from_id, env_id=env_id, num_envs=count, seed=seed,
                                     env_offset=start_idx + offset, **kwargs)
        shards.append(randomizer(env=make_env))
    return HipVecEnv(*shards)
