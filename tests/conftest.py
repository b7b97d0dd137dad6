import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib


@pytest.fixture(scope="session")
def engine():
    """One engine for the whole GPU session (sized for the largest test)."""
    from raft_amd import QuorumEngine
    e = QuorumEngine(max_groups=(1 << 20) + 1024, max_slots=16, max_log_terms=48 << 20)
    yield e
    e.close()
