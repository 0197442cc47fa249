"""Poly bank timing sweep (GPU box): voices x frames for every workgroup size (SMX_POLY_NT), with and without the
deferred fold (SMX_POLY_NO_DEFER) and with a grid override (SMX_POLY_GRID); one process per setting.
    python tools/explore_poly.py            -> one line per setting: us per block, 2^18 / 2^20 / 2^22 voices x 1/16/64 frames"""
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
        for nf in (1, 16, 64):
            for _ in range(200): pb.run_async(nf)
            pb.sync()
            best = 1e9
            for rep in range(5):
                pb.timer_start()
                for _ in range(100): pb.run_async(nf)
                best = min(best, pb.timer_stop() / 100)
            res["2^%d x %d" % (lg, nf)] = round(best * 1e3, 2)
        pb.close()
    print(json.dumps(res))

if __name__ == "__main__":
    if len(sys.argv) > 1: one(); sys.exit(0)
    for nt in ("256", "512", "1024"):
        for nd in ("", "1"):
            for g in ("0",) if nd else ("0", "512", "2048"):
                env = dict(os.environ); env["SMX_POLY_NT"] = nt; env["SMX_POLY_GRID"] = g
                if nd: env["SMX_POLY_NO_DEFER"] = "1"
                out = subprocess.run([sys.executable, __file__, "x"], env=env, capture_output=True, text=True)
                print("NT", nt, "no_defer" if nd else "deferred", "grid", g, out.stdout.strip(), out.stderr.strip()[-300:], flush=True)
