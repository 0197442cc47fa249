"""Scratch: small / mid banks at the JACK block length: chunk length and voices per lane of the direct kernel
(SMX_SAW_SMALL_VW, SMX_SAW_SMALL_TC, SMX_SAW_TC_CAP, SMX_SAW_GRID, SMX_SAW_NO_SLOTS)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth_tools_amd as sta
from synth_tools_amd import synthetic
tab = synthetic.note_inc_table(sta.lib().note_to_inc)
tag = " ".join("%s=%s" % (k[8:], os.environ[k]) for k in sorted(os.environ) if k.startswith("SMX_SAW_"))
line = "[%s]" % tag
for lg in [int(x) for x in os.environ.get("LGS", "14,16,18,20,22").split(",")]:
    n = 1 << lg
    inc, st = synthetic.saw_bank(n, 1, tab)
    b = sta.SawBank(n); b.load(inc, st)
    for nf in [int(x) for x in os.environ.get("NFS", "64").split(",")]:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.02:
            for _ in range(20): b.run_async(nf)
            b.sync()
        K = 300; b.timer_start()
        for _ in range(K): b.run_async(nf)
        ms = b.timer_stop() / K
        line += "  2^%d f%d %5.2f us" % (lg, nf, ms * 1e3)
    b.close()
print(line, flush=True)
