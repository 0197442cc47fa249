"""Scratch: does an idle gap before a timed run change the reading? (bench.py's secondary workloads
follow a 0.5 GB read-back and CPU work)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth_tools_amd as sta
from synth_tools_amd import synthetic
tab = synthetic.note_inc_table(sta.lib().note_to_inc)
n = 1 << 26
for seed in (1, 0x5EED0005):
    inc, st = synthetic.saw_bank(n, seed, tab)
    b = sta.SawBank(n); b.load(inc, st)
    for idle, warm, K in ((0, 20, 100), (1.0, 5, 50), (1.0, 50, 50), (0, 5, 50)):
        for _ in range(3): b.run_async(1)
        b.sync()
        time.sleep(idle)
        for _ in range(warm): b.run_async(16)
        b.sync(); b.timer_start()
        for _ in range(K): b.run_async(16)
        ms = b.timer_stop() / K
        print("seed %x idle %.1f warm %d K %d: %8.1f us" % (seed, idle, warm, K, ms * 1e3), flush=True)
    b.close()
