"""Stepping form pinned, timing over bank sizes and block lengths.  The inner loop is chosen per process:
SMX_SAW_NO_WIDE=1 = carry masks (rounds 1-2), default = 64-bit pairs.  Run both ways on one box and compare."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synth_tools_amd as sta
from synth_tools_amd import synthetic
tab = synthetic.note_inc_table(sta.lib().note_to_inc)
def t(bank, nf, steps):
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:            # clocks settle (tools/explore_settle.py)
        for _ in range(3): bank.run_async(nf)
        bank.sync()
    bank.timer_start()
    for _ in range(steps): bank.run_async(nf)
    ms = bank.timer_stop(); bank.sync()
    return ms / steps * 1e3
tag = "masks" if os.environ.get("SMX_SAW_NO_WIDE") else "wide "
for n in (1 << 24, 1 << 25, 1 << 26, 1 << 27):
    inc, st = synthetic.saw_bank(n, 0x5EED0001, tab)
    bank = sta.SawBank(n)
    bank.load(inc, st)
    bank.set_block_form(1)
    out = []
    for nf in (64, 128, 256, 1024):
        steps = max(8, int(3e10 / (n * nf)))
        out.append("%5d f %9.1f us" % (nf, t(bank, nf, steps)))
    print(tag, "%10d voices:" % n, " | ".join(out), flush=True)
    bank.close()
