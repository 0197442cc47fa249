"""Scratch: stepping carry form with 8 / 16 frames per chunk vs the direct form for 5..16-frame blocks.
SMX_SAW_CARRY_SHORT_MIN_LOG2=99 -> direct; =0 -> carry wherever allowed; SMX_SAW_CARRY_SHORT_MIN_FRAMES=5.."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth_tools_amd as sta
from synth_tools_amd import synthetic
tab = synthetic.note_inc_table(sta.lib().note_to_inc)
tag = " ".join("%s=%s" % (k[8:], os.environ[k]) for k in sorted(os.environ) if k.startswith("SMX_SAW_"))
for lg in [int(x) for x in os.environ.get("LGS", "24,25,26").split(",")]:
    n = 1 << lg
    inc, st = synthetic.saw_bank(n, 1, tab)
    b = sta.SawBank(n); b.load(inc, st)
    line = "%s 2^%d voices:" % (tag, lg)
    for nf in [int(x) for x in os.environ.get("NFS", "5,8,9,12,16").split(",")]:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.05:
            for _ in range(10): b.run_async(nf)
            b.sync()
        K = 100; b.timer_start()
        for _ in range(K): b.run_async(nf)
        ms = b.timer_stop() / K
        line += "  f%d %6.1f us (%.2f TB/s)" % (nf, ms * 1e3, 8.0 * n / ms / 1e9)
    print(line, flush=True)
    b.close()
