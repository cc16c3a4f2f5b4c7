import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# Robots that are not compiled in run on the generic kernels in the tests (those are what most of the
# run-time-model tests are about); the tests of gym_os2r_amd/jit.py switch the specialisation on themselves.
os.environ.setdefault("OS2R_JIT", "0")
# code objects built by those tests live in-tree (git-ignored and gpurun-ignored: the GPU box builds its own with its hipcc,
# ~7 s each; the library itself travels prebuilt)
KERNEL_CACHE = os.path.join(ROOT, ".kernel_cache")
# shards on streams (HipVecEnv(num_splits=...), bench.py --splits) want a hardware queue per shard stream; the HIP runtime reads
# this when it initialises -- in the test process that is long before the shard tests run (the suite ran them
# on shared queues, with the warning)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def native_libraries():
    """Build libos2r.so / _os2r_py.so / the oracle when a clean checkout runs the suite (the
    built files are git-ignored).  hipcc cross-compiles without a GPU; on the GPU box the
    libraries arrive prebuilt with the snapshot."""
    pkg = os.path.join(ROOT, "gym-os2r_amd")
    need = [os.path.join(pkg, "libos2r.so"), os.path.join(pkg, "_os2r_py.so"),
            os.path.join(ROOT, "oracle", "libos2r_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure), built on demand with gcc."""
    from oracle import oracle_py
    oracle_py.build()
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def pkg():
    import gym_os2r_amd
    return gym_os2r_amd
