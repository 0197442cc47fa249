"""Scratch: the 1-frame step of a 64 Mi-voice bank: saw_bank_kernel (1024 x 256 threads) vs saw_tick_kernel
(SMX_SAW_TICK_MAX_LOG2=27; SMX_SAW_TICK_GRID workgroups of 1024 threads), same box, same run."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth_tools_amd as sta
from synth_tools_amd import synthetic
tab = synthetic.note_inc_table(sta.lib().note_to_inc)
tag = " ".join("%s=%s" % (k[8:], os.environ[k]) for k in sorted(os.environ) if k.startswith("SMX_SAW_"))
n = 1 << int(os.environ.get('LG', '26'))
inc, st = synthetic.saw_bank(n, 0x5EED0005, tab)
b = sta.SawBank(n); b.load(inc, st)
line = "[%s] 2^%s" % (tag, os.environ.get("LG", "26"))
for nf in (1, 2, 4):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:
        for _ in range(10): b.run_async(nf)
        b.sync()
    best = 1e9
    for rep in range(3):
        K = 100; b.timer_start()
        for _ in range(K): b.run_async(nf)
        best = min(best, b.timer_stop() / K)
    line += "  f%d %6.2f us (%.2f TB/s)" % (nf, best * 1e3, 8.0 * n / best / 1e9)
    iso = []
    for rep in range(40):                       # one launch at a time: no overlap with a predecessor's tail
        b.timer_start(); b.run_async(nf); iso.append(b.timer_stop())
    iso.sort()
    line += " [isolated median %.2f min %.2f]" % (iso[len(iso) // 2] * 1e3, iso[0] * 1e3)
print(line, flush=True)
b.close()
