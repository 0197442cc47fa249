"""Scratch: kernel rates of the PDM / PWM / poly banks."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth_tools_amd as sta
from synth_tools_amd import synthetic
tab = synthetic.note_inc_table(sta.lib().note_to_inc)

for n, nt in ((1 << 20, 4096), (1 << 22, 1024), (1 << 16, 16384)):
    sp, ac = synthetic.pdm_bank(n, 3)
    p = sta.PdmBank(n); p.load(sp, ac)
    for wd in (False, True):
        if wd: p.tick_n(64, synthetic.dither_stream(64, 7, 0x0FFFFFFF), want_bits=False)
        p.tick_n_async(nt, wd); p.sync(); p.timer_start()
        for _ in range(5): p.tick_n_async(nt, wd)
        ms = p.timer_stop() / 5
        print("pdm n=%8d nt=%6d dither=%d: %8.4f ms %9.1f G ch-ticks/s out %7.1f GB/s" % (n, nt, wd, ms, n*nt/ms/1e6, n*nt/8/ms/1e6), flush=True)
    p.close()

for order in (1, 2, 4):
    for n, nt in ((1 << 20, 1024), (1 << 16, 8192)):
        b = sta.PwmBank(n, order=order)
        r = synthetic.splitmix64(5, n)
        b.load(setpoint=(r >> np.uint64(32)).astype(np.uint32))
        for wd in (False, True):
            if wd: b.tick_n(8, synthetic.dither_stream(8, 7, 0x3FF), want_duty=False)
            b.tick_n_async(nt, wd); b.sync(); b.timer_start()
            for _ in range(5): b.tick_n_async(nt, wd)
            ms = b.timer_stop() / 5
            print("pwm order=%d n=%8d nt=%6d dither=%d: %8.4f ms %9.1f G ch-ticks/s out %7.1f GB/s" % (order, n, nt, wd, ms, n*nt/ms/1e6, n*nt/ms/1e6), flush=True)
        b.close()

for n in (1 << 18, 1 << 22):
    pb = sta.PolyBank(n); pb.load(**synthetic.poly_bank(n, 4, tab))
    for B in (1, 16, 64):
        for _ in range(3): pb.run_async(B)
        pb.sync(); pb.timer_start()
        for _ in range(20): pb.run_async(B)
        ms = pb.timer_stop() / 20
        print("poly n=%8d B=%3d: %8.4f ms %9.1f Gs/s alg %7.1f GB/s" % (n, B, ms, n*B/ms/1e6, n*60/ms/1e6), flush=True)
    pb.close()
