"""Scratch: carry-out PDM bank and PWM bank in the tick regime (few ticks per launch, big banks):
algorithmic bytes 8*N + T*N/8 (PDM: the accumulator is lazy since round 3, nothing is written back), 52*N + T*N (PWM order 2) against the HBM peak."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth_tools_amd as sta
from synth_tools_amd import synthetic

for lg in (20, 26):
    n = 1 << lg
    sp, ac = synthetic.pdm_bank(n, 3)
    p = sta.PdmBank(n); p.load(sp, ac)
    for nt in (1, 2, 4, 8, 64, 4096):
        if n * nt // 8 > (1 << 30): continue
        for streams in (False, True):
            if streams and nt % 32: continue
            f = (lambda: p.tick_n_streams_async(nt, False)) if streams else (lambda: p.tick_n_async(nt, False))
            f(); p.sync(); p.timer_start()
            for _ in range(10): f()
            ms = p.timer_stop() / 10
            alg = 8.0 * n + nt * n / 8
            print("pdm%s n=2^%d nt=%4d: %8.1f us %9.1f G ch-ticks/s  alg %7.1f GB/s (%.0f%% HBM)" % (
                "-streams" if streams else "        ", lg, nt, ms * 1e3, n * nt / ms / 1e6, alg / ms / 1e6, alg / ms / 1e6 / 80), flush=True)
    p.close()

for lg in (20, 24):
    n = 1 << lg
    b = sta.PwmBank(n, order=2)
    r = synthetic.splitmix64(5, n)
    b.load(setpoint=(r >> np.uint64(32)).astype(np.uint32))
    for nt in (1, 8, 64):
        b.tick_n(8, synthetic.dither_stream(8, 7, 0x3FF), want_duty=False)
        b.tick_n_async(nt, True); b.sync(); b.timer_start()
        for _ in range(10): b.tick_n_async(nt, True)
        ms = b.timer_stop() / 10
        alg = 52.0 * n + nt * n
        print("pwm2 n=2^%d nt=%4d: %8.1f us %9.1f G ch-ticks/s  alg %7.1f GB/s (%.0f%% HBM)" % (
            lg, nt, ms * 1e3, n * nt / ms / 1e6, alg / ms / 1e6, alg / ms / 1e6 / 80), flush=True)
    b.close()
