"""How long does the 1-frame step of the headline bank take to reach its steady rate in a fresh process?
Windows of 200 back-to-back steps (HIP events on the bank's stream), one after the other, for ~1.5 s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth_tools_amd as sta
from synth_tools_amd import synthetic
tab = synthetic.note_inc_table(sta.lib().note_to_inc)
n = 1 << 26
inc, st = synthetic.saw_bank(n, 0x5EED0005, tab)
bank = sta.SawBank(n)
bank.load(inc, st)
out = []
for w in range(100):
    bank.timer_start()
    for _ in range(200): bank.run_async(1)
    out.append(bank.timer_stop() / 200 * 1e3)
print(" ".join("%.1f" % x for x in out))
