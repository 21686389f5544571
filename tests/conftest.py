import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "soak: long GPU runs (bank-size soaks, the scheduler-variant builds); part of -m gpu, "
                                       "left out by -m \"gpu and not soak\"")


@pytest.fixture(scope="session")
def built():
    """Make sure libspangpu.so and the oracle exist (cross-compiles without a GPU)."""
    import __graft_entry__ as g
    from spandsp_amd import engine
    if not os.path.exists(engine.LIB_PATH):
        g.build()
    import oracle
    if not os.path.exists(oracle.ORACLE_SO):
        oracle.build()
    return True
