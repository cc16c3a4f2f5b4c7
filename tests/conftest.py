import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
