"""Chunk length of the direct form for 2^20 .. 2^23-voice banks at 64 frames (SMX_SAW_TC_CAP: one process per value):
shorter chunks put more workgroups on the chip (a 1 Mi-voice bank is 1024 rows: 4 waves per SIMD with 64-frame chunks).
    python tools/explore_tc_cap.py   -> us per 64-frame block in a stream of un-fetched blocks"""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def one():
    import numpy as np
    import synth_tools_amd as sta
    from synth_tools_amd import synthetic
    tab = synthetic.note_inc_table(sta.lib().note_to_inc)
    res = {}
    for lg in (20, 21, 22, 23):
        n = 1 << lg
        inc, st = synthetic.saw_bank(n, 0x5EED0005, tab)
        b = sta.SawBank(n); b.load(inc, st)
        for nf in (64, 32):
            for _ in range(300): b.run_async(nf)
            b.sync()
            best = 1e9
            for rep in range(5):
                b.timer_start()
                for _ in range(200): b.run_async(nf)
                best = min(best, b.timer_stop() / 200)
            res["2^%d x %d" % (lg, nf)] = round(best * 1e3, 2)
        b.close()
    print(json.dumps(res))

if __name__ == "__main__":
    if len(sys.argv) > 1: one(); sys.exit(0)
    for cap in ("64", "32", "16"):
        env = dict(os.environ); env["SMX_SAW_TC_CAP"] = cap
        out = subprocess.run([sys.executable, __file__, "x"], env=env, capture_output=True, text=True)
        print("TC_CAP", cap, out.stdout.strip(), out.stderr.strip()[-300:], flush=True)
