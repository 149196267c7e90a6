import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from tests import oracle_lib

    return oracle_lib.load()


@pytest.fixture(scope="session")
def gg():
    """One gg_ctx on cuda:0 for the whole GPU session (fails loudly if libgg.so is missing)."""
    import duckdb_pgq_amd as pkg

    g = pkg.GG(0)
    yield g
    g.close()
