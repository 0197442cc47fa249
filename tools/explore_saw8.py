"""Scratch: direct formulation with bus slots, 64 Mi and 16 Mi voices x 8/16 frames (prefetch depth A/B)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth_tools_amd as sta
from synth_tools_amd import synthetic
tab = synthetic.note_inc_table(sta.lib().note_to_inc)
for lg in (24, 26):
    n = 1 << lg
    inc, st = synthetic.saw_bank(n, 1, tab)
    b = sta.SawBank(n); b.load(inc, st)
    line = "n=2^%d" % lg
    for B in (8, 16, 5):
        for _ in range(20): b.run_async(B)
        b.sync(); K = 100; b.timer_start()
        for _ in range(K): b.run_async(B)
        ms = b.timer_stop() / K
        line += "  B=%d %8.1f us %7.0f Gs/s %5.2f TB/s" % (B, ms * 1e3, n * B / ms / 1e6, n * 8 / ms / 1e9)
    print(line, flush=True)
    b.close()
