"""Scratch: default grid heuristics across bank sizes and block lengths."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth_tools_amd as sta
from synth_tools_amd import synthetic
tab = synthetic.note_inc_table(sta.lib().note_to_inc)
for n in (1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 24, 1 << 26):
    inc, st = synthetic.saw_bank(n, 1, tab)
    b = sta.SawBank(n); b.load(inc, st)
    for B in (1, 16, 64, 1024):
        if n * B > (1 << 35): continue
        for _ in range(5): b.run_async(B)
        b.sync(); K = 50; b.timer_start()
        for _ in range(K): b.run_async(B)
        ms = b.timer_stop() / K
        print("n=%9d B=%5d %8.4f ms %9.1f Gs/s" % (n, B, ms, n*B/ms/1e6), flush=True)
    b.close()
