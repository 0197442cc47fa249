"""Scratch: where the carry formulation (stepping / events, AUTO) overtakes the direct one at 64 frames
(SMX_SAW_CARRY_MIN_LOG2 = log2 of the voice-samples per launch from which it is used)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth_tools_amd as sta
from synth_tools_amd import synthetic
tab = synthetic.note_inc_table(sta.lib().note_to_inc)
line = "min_log2=%s form=%s:" % (os.environ.get("SMX_SAW_CARRY_MIN_LOG2", "31"), os.environ.get("FORM", "0"))
for lg in (20, 21, 22, 23, 24, 25):
    n = 1 << lg
    inc, st = synthetic.saw_bank(n, 1, tab)
    b = sta.SawBank(n); b.set_block_form(int(os.environ.get("FORM", "0"))); b.load(inc, st)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.02:
        for _ in range(10): b.run_async(64)
        b.sync()
    K = 200; b.timer_start()
    for _ in range(K): b.run_async(64)
    ms = b.timer_stop() / K
    line += "  2^%d %6.1f us" % (lg, ms * 1e3)
    b.close()
print(line, flush=True)
