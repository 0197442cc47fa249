"""Does the speed of the headline step depend on WHERE the bank lies in HBM?  Several banks of the same content created
in one process (the earlier ones kept alive, so that each lands somewhere else), each timed over 400 steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth_tools_amd as sta
from synth_tools_amd import synthetic
tab = synthetic.note_inc_table(sta.lib().note_to_inc)
n = 1 << 26
inc, st = synthetic.saw_bank(n, 0x5EED0005, tab)
def t(bank):
    for _ in range(50): bank.run_async(1)
    bank.sync(); bank.timer_start()
    for _ in range(400): bank.run_async(1)
    ms = bank.timer_stop(); bank.sync()
    return ms / 400 * 1e3
banks = []
for k in range(8):
    b = sta.SawBank(n); b.load(inc, st); banks.append(b)
    print("bank %d (all earlier ones alive): %.2f us" % (k, t(b)), flush=True)
print("again, in creation order:", " ".join("%.2f" % t(b) for b in banks), flush=True)
for b in banks: b.close()
b = sta.SawBank(n); b.load(inc, st)
print("after freeing all, a new bank: %.2f us" % t(b), flush=True)
