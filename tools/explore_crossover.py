"""Scratch: from how many voice-samples per launch does the carry path (AUTO: events on a piano bank) beat the
direct form at 64 frames?  SMX_SAW_CARRY_MIN_LOG2 = 31 (never for these sizes) vs 28 (always)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth_tools_amd as sta
from synth_tools_amd import synthetic
tab = synthetic.note_inc_table(sta.lib().note_to_inc)
form = int(os.environ.get("SMX_EXPLORE_FORM", "0"))          # 0 AUTO, 1 STEPPING pinned
line = "carry_min_log2=%s form=%d:" % (os.environ.get("SMX_SAW_CARRY_MIN_LOG2", "30"), form)
for lg in (21, 22, 23, 24, 25):
    n = 1 << lg
    inc, st = synthetic.saw_bank(n, 1, tab)
    b = sta.SawBank(n); b.load(inc, st)
    b.set_block_form(form)
    for nf in (64, 128):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.04:
            for _ in range(10): b.run_async(nf)
            b.sync()
        K = 100; b.timer_start()
        for _ in range(K): b.run_async(nf)
        line += "  2^%d f%d %6.1f us" % (lg, nf, b.timer_stop() / K * 1e3)
    b.close()
print(line, flush=True)
