"""Scratch sweep: grid cap x frames on a 64 Mi-voice bank (one process per cap: env is read once)."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import numpy as np
    import synth_tools_amd as sta
    from synth_tools_amd import synthetic
    n = 1 << 26
    tab = synthetic.note_inc_table(sta.lib().note_to_inc)
    inc, st = synthetic.saw_bank(n, 1, tab)
    b = sta.SawBank(n); b.load(inc, st)
    for B in (1, 4, 16, 32, 64):
        for _ in range(3): b.run_async(B)
        b.sync(); K = 30; b.timer_start()
        for _ in range(K): b.run_async(B)
        ms = b.timer_stop() / K
        print("cap=%s B=%3d %7.4f ms %9.1f Gs/s alg %7.1f GB/s" % (sys.argv[1], B, ms, n*B/ms/1e6, (n*12+B*4)/ms/1e6), flush=True)
else:
    for cap in ("256", "512", "768", "1024", "2048"):
        subprocess.run([sys.executable, __file__, cap], env=dict(os.environ, SMX_SAW_GRID=cap))
