"""Scratch sweep: saw-bank kernel time over (voices, frames).  Not the bench."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth_tools_amd as sta
from synth_tools_amd import synthetic

tab = synthetic.note_inc_table(sta.lib().note_to_inc)
for n in (1 << 16, 1 << 20, 1 << 24, 1 << 26):
    inc, st = synthetic.saw_bank(n, 1, tab)
    b = sta.SawBank(n); b.load(inc, st)
    for B in (1, 4, 16, 64, 256, 1024):
        if n * B > (1 << 37): continue
        for _ in range(3): b.run_async(B)
        b.sync()
        K = 20
        b.timer_start()
        for _ in range(K): b.run_async(B)
        ms = b.timer_stop() / K
        vs = n * B / (ms * 1e-3)
        gbs = (n * 12 + B * 4) / (ms * 1e-3) / 1e9
        print("n=%9d B=%5d  %8.3f ms/step  %9.1f Gsamples/s  alg %7.1f GB/s" % (n, B, ms, vs / 1e9, gbs), flush=True)
    b.close()
