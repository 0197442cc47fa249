"""The library's A/B switches (tools/README.md) select older or alternative code paths that are documented as
bit-exact: each one set in a process of its own, the same blocks against the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
SWITCHES = ["", "SMX_SAW_NO_WIDE", "SMX_SAW_NO_DEFER", "SMX_SAW_NO_SLOTS", "SMX_SAW_NO_CARRY", "SMX_SAW_NO_LONG_EVENTS", "SMX_NO_PUBLISH",
            "SMX_POLY_NO_DEFER", "SMX_SAW_EVENTS_256", "SMX_PDM_NO_FEWTICKS", "SMX_BANK_TWO_ALLOCS", "SMX_SAW_NO_SHORT_EVENTS",
            "SMX_SAW_NO_EVENTS_128"]


@pytest.mark.parametrize("switch", SWITCHES)
def test_switch_keeps_the_bus_bit_exact(smx, switch):
    env = dict(os.environ)
    for s in SWITCHES:
        env.pop(s, None)
    if switch:
        env[switch] = "1"
    p = subprocess.run([sys.executable, os.path.join(HERE, "env_switch_worker.py")], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stdout.strip().startswith("ok"), p.stdout[-500:]
