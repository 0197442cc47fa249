"""Scratch: BASELINE config 2 (65 536 voices) in long launches: direct formulation vs the carry / event forms."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth_tools_amd as sta
from synth_tools_amd import synthetic
tab = synthetic.note_inc_table(sta.lib().note_to_inc)
line = "small_long_log2=%s:" % os.environ.get("SMX_SAW_SMALL_LONG_LOG2", "27")
for n in (1 << 16, 1 << 18):
    inc, st = synthetic.saw_bank(n, 0x5EED0002, tab)
    b = sta.SawBank(n); b.load(inc, st)
    for B in (1024, 4096, 16384):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.02:
            for _ in range(5): b.run_async(B)
            b.sync()
        K = 50; b.timer_start()
        for _ in range(K): b.run_async(B)
        ms = b.timer_stop() / K
        line += "  n=2^%d B=%d %6.1f us %6.1f Ts/s" % (n.bit_length() - 1, B, ms * 1e3, n * B / ms / 1e9)
    b.close()
print(line, flush=True)
