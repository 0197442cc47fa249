"""Scratch: c3 (1 Mi channels x 4096 ticks) tick-major PDM kernel, one vs two channels per lane (SMX_PDM_V1=1),
grid sizes (SMX_PDM_GRID), and the few-tick regime of a big bank."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth_tools_amd as sta
from synth_tools_amd import synthetic
tag = "v1=%s grid=%s" % (os.environ.get("SMX_PDM_V1", "0"), os.environ.get("SMX_PDM_GRID", "dflt"))
def run(n, nt, with_d, reps):
    sp, ac = synthetic.pdm_bank(n, 0x5EED0003)
    p = sta.PdmBank(n); p.load(sp, ac)
    if with_d: p.tick_n(nt, synthetic.dither_stream(nt, 7, 0x0FFFFFFF), want_bits=False)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.03:
        p.tick_n_async(nt, with_d); p.sync()
    p.timer_start()
    for _ in range(reps): p.tick_n_async(nt, with_d)
    ms = p.timer_stop() / reps
    p.close()
    return ms
line = tag + ":"
for n, nt, reps in ((1 << 20, 4096, 20), (1 << 20, 100, 50), (1 << 26, 1, 20), (1 << 26, 8, 20), (1 << 24, 32, 20), (5 << 10, 4096, 20)):
    for d in (False, True):
        line += "  %dx%d%s %.1f us" % (n, nt, "+d" if d else "", run(n, nt, d, reps) * 1e3)
print(line, flush=True)
