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
    """The CPU oracle (checker only)."""
    import oracle
    return oracle.load()


@pytest.fixture(scope="session")
def smx():
    """The product: ctypes binding of libsynth_mi355x.so."""
    import synth_tools_amd
    synth_tools_amd.lib()
    return synth_tools_amd


@pytest.fixture(scope="session")
def inc_table(smx):
    from synth_tools_amd import synthetic
    return synthetic.note_inc_table(smx.lib().note_to_inc)
