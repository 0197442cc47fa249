"""Scratch: tick-mode (1..4 frames) rates vs SMX_SAW_TICK_GRID."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import numpy as np
    import synth_tools_amd as sta
    from synth_tools_amd import synthetic
    tab = synthetic.note_inc_table(sta.lib().note_to_inc)
    for n in (1 << 22, 1 << 24, 1 << 26):
        inc, st = synthetic.saw_bank(n, 1, tab)
        b = sta.SawBank(n); b.load(inc, st)
        for B in (1, 4):
            for _ in range(5): b.run_async(B)
            b.sync(); K = 50; b.timer_start()
            for _ in range(K): b.run_async(B)
            ms = b.timer_stop() / K
            print("tickgrid=%4s n=%9d B=%d %8.4f ms %8.1f Gs/s %7.1f GB/s" % (sys.argv[1], n, B, ms, n*B/ms/1e6, n*8/ms/1e6), flush=True)
        b.close()
else:
    for g in ("192", "256", "512"):
        subprocess.run([sys.executable, __file__, g], env=dict(os.environ, SMX_SAW_TICK_GRID=g))
