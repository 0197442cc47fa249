"""Scratch: carry formulation vs direct formulation at B in (33..1024) on big banks."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import numpy as np
    import synth_tools_amd as sta
    from synth_tools_amd import synthetic
    tab = synthetic.note_inc_table(sta.lib().note_to_inc)
    for n in (1 << 20, 1 << 22, 1 << 24):
        inc, st = synthetic.saw_bank(n, 1, tab)
        b = sta.SawBank(n); b.load(inc, st)
        for B in (64, 128, 1024):
            for _ in range(3): b.run_async(B)
            b.sync(); K = 20; b.timer_start()
            for _ in range(K): b.run_async(B)
            ms = b.timer_stop() / K
            print("%s n=%9d B=%5d %8.4f ms %9.1f Gs/s alg %7.1f GB/s" % (sys.argv[1], n, B, ms, n*B/ms/1e6, (n*12+B*4)/ms/1e6), flush=True)
        b.close()
else:
    subprocess.run([sys.executable, __file__, "direct"], env=dict(os.environ, SMX_SAW_NO_CARRY="1"))
