"""PDM bank, 64 Mi channels, 1 and 2 ticks per launch (the read-stream kernel; SMX_PDM_NO_FEWTICKS=1: the tile
kernel), before and after a read-back (which materialises the lazy accumulators: the arrays are rewritten)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth_tools_amd as sta
from synth_tools_amd import synthetic
n = 1 << 26
sp, ac = synthetic.pdm_bank(n, 3)
p = sta.PdmBank(n); p.load(sp, ac)

def t(nt, reps=100):
    p.tick_n_async(nt, False); p.sync(); p.timer_start()
    for _ in range(reps): p.tick_n_async(nt, False)
    return round(p.timer_stop() / reps * 1e3, 1)

print("fresh bank:          1 tick", t(1), "us   2 ticks", t(2), "us", flush=True)
p.read()
print("after read():        1 tick", t(1), "us   2 ticks", t(2), "us", flush=True)
p.tick_n(1)
print("after tick_n(1):     1 tick", t(1), "us   2 ticks", t(2), "us", flush=True)
p.tick_n(2)
print("after tick_n(2):     1 tick", t(1), "us   2 ticks", t(2), "us", flush=True)
time.sleep(1.0)
print("after 1 s idle:      1 tick", t(1), "us   2 ticks", t(2), "us", flush=True)
