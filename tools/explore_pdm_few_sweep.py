"""PDM bank, 64 Mi channels, 1..8 ticks per launch: the read-stream kernel (pdm_fewticks_kernel, admitted up to
SMX_PDM_FEWTICKS_MAX ticks) against the tile kernel (SMX_PDM_NO_FEWTICKS=1), one process each; bits of the last launch
checked against the closed form (dither 0: pulse at tick t iff accu0 + (T+t+1)*sp wrapped in that step).
    python tools/explore_pdm_few_sweep.py"""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one():
    import numpy as np
    import synth_tools_amd as sta
    from synth_tools_amd import synthetic
    n = 1 << 26
    sp, ac = synthetic.pdm_bank(n, 3)
    p = sta.PdmBank(n); p.load(sp, ac)
    res = {}
    T = 0
    for nt in (1, 2, 3, 4, 6, 8):
        for _ in range(20): p.tick_n_async(nt, False)
        T += 20 * nt
        p.sync()
        best = 1e9
        for rep in range(4):
            p.timer_start()
            for _ in range(50): p.tick_n_async(nt, False)
            best = min(best, p.timer_stop() / 50)
            T += 50 * nt
        bits = p.tick_n(nt)                                            # checked: words of the first and last 4096 channels
        ok = True
        for lo in (0, n - 4096):
            a = (ac[lo:lo + 4096].astype(np.uint64) + np.uint64(T) * sp[lo:lo + 4096].astype(np.uint64)) & np.uint64(0xFFFFFFFF)
            for t in range(nt):
                nxt = a + sp[lo:lo + 4096].astype(np.uint64)
                carry = (nxt >> np.uint64(32)).astype(np.uint8)
                a = nxt & np.uint64(0xFFFFFFFF)
                got = ((bits[t, lo // 32:(lo + 4096) // 32, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(-1).astype(np.uint8)
                ok = ok and bool(np.array_equal(got, carry))
        T += nt
        alg = 8.0 * n + nt * n / 8
        res[nt] = (round(best * 1e3, 1), "%.0f%%" % (alg / (best * 1e-3) / 8e12 * 100), "ok" if ok else "WRONG")
    print(json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 1: one(); sys.exit(0)
    for name, env in (("few<=8", {"SMX_PDM_FEWTICKS_MAX": "8"}), ("tile", {"SMX_PDM_NO_FEWTICKS": "1"}), ("default", {})):
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, __file__, "x"], env=e, capture_output=True, text=True)
        print("%-8s" % name, out.stdout.strip(), out.stderr.strip()[-400:], flush=True)
