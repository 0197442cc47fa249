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


@pytest.fixture(autouse=True, scope="session")
def _soak_seed():
    """SMX_SOAK_SEED=<int>: every numpy generator a test seeds itself is re-seeded with (seed XOR this value), so
    that the parity tests (HIP path vs oracle on the same inputs) run on other data than the committed seeds give.
    Off by default; tests that compare with committed fixtures do not draw their inputs from such generators."""
    mix = os.environ.get("SMX_SOAK_SEED")
    if not mix:
        yield
        return
    import numpy as np
    mix = int(mix, 0)
    orig = np.random.default_rng

    def seeded(seed=None):
        return orig(seed if not isinstance(seed, int) else seed ^ mix)
    np.random.default_rng = seeded
    try:
        yield
    finally:
        np.random.default_rng = orig
