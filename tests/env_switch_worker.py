"""One process with one of the library's A/B switches set (test helper for tests/test_env_switches_gpu.py): blocks
that reach the code path the switch selects, compared with the oracle.  Prints "ok <checks>"."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import oracle
    import synth_tools_amd as sta
    from synth_tools_amd import synthetic
    orc = oracle.load()
    tab = synthetic.note_inc_table(sta.lib().note_to_inc)
    checks = 0
    # (voices, block lengths): direct slot launches, tick launches, carry forms (stepping and events), small banks
    for n, frames, form in (((1 << 20) + 3, (8, 16, 64, 64, 3, 100, 32), 0),
                            (1 << 23, (64, 128, 64), 1),               # carry, stepping pinned
                            (1 << 23, (64, 300, 64), 2),               # carry, events pinned (incl. a long launch)
                            (70000, (64, 1, 17, 200), 0)):
        inc, st = synthetic.saw_bank(n, 0x5EED0E00 + n % 977, tab, active_fraction=0.9)
        bank = sta.SawBank(n)
        bank.load(inc, st)
        bank.set_block_form(form)
        for k, nf in enumerate(frames):
            bank.run_async(nf)
            want, _ = oracle.synth_run(orc, inc, st, nf, want_vec=False)
            if k % 2 == 1 or k == len(frames) - 1:          # un-fetched blocks in between
                bus, _ = bank.fetch(nf)
                assert np.array_equal(bus, want), (n, nf, form)
                checks += 1
        assert np.array_equal(bank.read()[1], st), (n, "phases")
        bank.close()
    print("ok", checks)


if __name__ == "__main__":
    main()
