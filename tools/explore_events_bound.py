"""Scratch: how slow can the wrap-event form get on banks that the device-side rule ADMITS (largest increment
below 6.5 * 2^26 and mean increment <= 2^27, i.e. <= 2 wraps per voice and 64 frames)?  Forced forms, 64 Mi voices x 64 frames."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth_tools_amd as sta
from synth_tools_amd import synthetic
tab = synthetic.note_inc_table(sta.lib().note_to_inc)
n = 1 << int(os.environ.get("LG", "26"))
r = synthetic.splitmix64(99, n)
u = (r >> np.uint64(11)).astype(np.float64) / float(1 << 53)
W = lambda w: np.uint32(int(w * (1 << 26)))          # increment that wraps w times per 64 frames
banks = {
    "piano range": synthetic.saw_bank(n, 1, tab)[0],
    "all at 1 wrap": np.full(n, W(1.0), np.uint32),
    "all at 2 wraps (edge of the rule)": np.full(n, W(2.0) - 1, np.uint32),
    "31% at 6.4 wraps, rest inc=1": np.where(u < 0.31, W(6.4), 1).astype(np.uint32),
    "50% at 4 wraps, rest inc=1": np.where(u < 0.5, W(3.99), 1).astype(np.uint32),
    "every 4th voice at 6.4 wraps, rest inc=1": np.where(np.arange(n) % 4 == 0, W(6.4), 1).astype(np.uint32),
    "one voice per wave-row at 6.4 wraps, rest at 1.9": np.where(np.arange(n) % 256 == 0, W(6.4), W(1.9)).astype(np.uint32),
    "2% at 6.4 wraps, rest at 1.9 wraps": np.where(u < 0.02, W(6.4), W(1.9)).astype(np.uint32),
    "uniform 0..4 wraps": (u * 4.0 * (1 << 26)).astype(np.uint32) + 1,
}
st = (r >> np.uint64(32)).astype(np.uint32)
b = sta.SawBank(n)
for name, inc in banks.items():
    mean_w = inc.astype(np.float64).mean() / (1 << 26)
    line = "%-52s mean %.2f wraps max %.2f:" % (name, mean_w, inc.max() / (1 << 26))
    for form, fname in ((1, "stepping"), (2, "events"), (0, "auto")):
        b.set_block_form(form)
        b.load(inc, st)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.04:
            for _ in range(5): b.run_async(64)
            b.sync()
        K = 40; b.timer_start()
        for _ in range(K): b.run_async(64)
        line += "  %s %6.1f us" % (fname, b.timer_stop() / K * 1e3)
    print(line, flush=True)
b.close()
