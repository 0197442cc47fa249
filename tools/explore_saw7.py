"""Scratch: carry formulation, 64 Mi voices x 32/64/1024 frames (run with SMX_SAW_CARRY_GRID=...)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth_tools_amd as sta
from synth_tools_amd import synthetic
tab = synthetic.note_inc_table(sta.lib().note_to_inc)
n = 1 << 26
inc, st = synthetic.saw_bank(n, 1, tab)
b = sta.SawBank(n); b.set_block_form(int(os.environ.get("FORM", "0"))); b.load(inc, st)
line = "form=%s grid=%s" % (os.environ.get("FORM", "0"), os.environ.get("SMX_SAW_CARRY_GRID", "default"))
import time
for B in (64, 1024):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.03:          # steady state (clock ramp)
        for _ in range(10): b.run_async(B)
        b.sync()
    K = 100 if B < 1024 else 10; b.timer_start()
    for _ in range(K): b.run_async(B)
    ms = b.timer_stop() / K
    line += "  B=%d %8.1f us %7.0f Gs/s" % (B, ms * 1e3, n * B / ms / 1e6)
print(line, flush=True)
b.close()
