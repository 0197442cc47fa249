"""Scratch: how long after an idle gap do timings of the 16-frame block settle?  (bench.py settles 10 ms.)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth_tools_amd as sta
from synth_tools_amd import synthetic
tab = synthetic.note_inc_table(sta.lib().note_to_inc)
n = 1 << 26
inc, st = synthetic.saw_bank(n, 0x5EED0005, tab)
b = sta.SawBank(n); b.load(inc, st)
for nf in (16, 64, 1):
    for gap in (0.0, 1.0, 3.0):
        b.sync(); time.sleep(gap)
        out = []
        t0 = time.perf_counter()
        for k in range(12):
            b.timer_start()
            for _ in range(20): b.run_async(nf)
            out.append("%.1f" % (b.timer_stop() / 20 * 1e3))
        print("f%d after %.0f s idle: us per step in consecutive batches of 20 (total %.1f ms): %s" % (nf, gap, (time.perf_counter() - t0) * 1e3, " ".join(out)), flush=True)
b.close()
