"""Scratch: mid-size banks at 64 frames vs persistent grid size (env SMX_SAW_GRID read once per process)."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import numpy as np
    import synth_tools_amd as sta
    from synth_tools_amd import synthetic
    tab = synthetic.note_inc_table(sta.lib().note_to_inc)
    for n in (1 << 16, 1 << 18, 1 << 20, 1 << 22):
        inc, st = synthetic.saw_bank(n, 1, tab)
        b = sta.SawBank(n); b.load(inc, st)
        for B in (16, 64):
            for _ in range(5): b.run_async(B)
            b.sync(); K = 100; b.timer_start()
            for _ in range(K): b.run_async(B)
            ms = b.timer_stop() / K
            print("grid=%5s n=%8d B=%3d %8.4f ms %9.1f Gs/s" % (sys.argv[1], n, B, ms, n*B/ms/1e6), flush=True)
        b.close()
else:
    for g in ("128", "256", "512", "1024", "2048"):
        subprocess.run([sys.executable, __file__, g], env=dict(os.environ, SMX_SAW_GRID=g))
