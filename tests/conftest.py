import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the extension prints a backtrace if the process dies of a signal (gg_duckdb_extension.cpp): the GPU boxes have no
# debugger, and a fault inside the reference's worker threads would otherwise leave nothing but an exit code
os.environ.setdefault("GG_CRASH_TRACE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from tests import oracle_lib

    return oracle_lib.load()


@pytest.fixture(scope="session")
def _gg_session():
    """One gg_ctx on cuda:0 for the whole GPU session (fails loudly if libgg.so is missing)."""
    import duckdb_pgq_amd as pkg

    g = pkg.GG(0)
    yield g
    g.close()


@pytest.fixture
def gg(_gg_session):
    """The session's context with every knob at its default: whatever a test switched (edge rowids, rank mode,
    forced builds, scan faults, launch splitting, profiling) and whatever it left staged is undone after it,
    also when it failed — a test must not depend on the tests that ran before it."""
    g = _gg_session
    g.debug_reset()
    yield g
    g.debug_reset()
    g.profile(False)
    g.staging_clear()
