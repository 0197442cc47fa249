"""Poly bank timing sweep (GPU box): voices x frames, grid override via SMX_POLY_GRID (one process each)."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def one():
    import numpy as np
    import synth_tools_amd as sta
    from synth_tools_amd import synthetic
    tab = synthetic.note_inc_table(sta.lib().note_to_inc)
    res = {}
    for lg in (18, 20, 22):
        n = 1 << lg
        pb = sta.PolyBank(n)
        pb.load(**synthetic.poly_bank(n, 0x5EED0004, tab))
        for nf in (1, 4, 16, 64):
            for _ in range(3): pb.run_async(nf)
            pb.sync(); pb.timer_start()
            for _ in range(30): pb.run_async(nf)
            ms = pb.timer_stop() / 30
            res["2^%d x %d" % (lg, nf)] = round(ms * 1e3, 1)
        pb.close()
    print(json.dumps(res))

if __name__ == "__main__":
    if len(sys.argv) > 1: one(); sys.exit(0)
    for g in ("0", "256", "512", "1024", "2048", "4096"):
        env = dict(os.environ); env["SMX_POLY_GRID"] = g
        out = subprocess.run([sys.executable, __file__, "x"], env=env, capture_output=True, text=True)
        print("grid", g, out.stdout.strip(), out.stderr.strip()[-300:], flush=True)
