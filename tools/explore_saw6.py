"""Scratch: mid-size banks x 16..64 frames (direct formulation below 2^31 voice-samples): how far from
the 15.7 Ts/s issue bound, and what the end-of-kernel bus atomics cost (grid override)."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def one():
    import synth_tools_amd as sta
    from synth_tools_amd import synthetic
    tab = synthetic.note_inc_table(sta.lib().note_to_inc)
    for lg in (18, 20, 21, 22, 23, 24):
        n = 1 << lg
        inc, st = synthetic.saw_bank(n, 1, tab)
        b = sta.SawBank(n); b.load(inc, st)
        line = "n=2^%d" % lg
        for B in (8, 16, 32, 64):
            for _ in range(5): b.run_async(B)
            b.sync(); K = 50; b.timer_start()
            for _ in range(K): b.run_async(B)
            ms = b.timer_stop() / K
            line += "  B=%d %7.1f us %7.0f Gs/s" % (B, ms * 1e3, n * B / ms / 1e6)
        print(line, flush=True)
        b.close()

if __name__ == "__main__":
    if len(sys.argv) > 1: one(); sys.exit(0)
    for g in ("", "512", "1024", "4096", "noslots"):
        env = dict(os.environ)
        if g == "noslots": env["SMX_SAW_NO_SLOTS"] = "1"
        elif g: env["SMX_SAW_SLOT_GRID"] = g
        print("SMX_SAW_SLOT_GRID=%s" % (g or "default"), flush=True)
        out = subprocess.run([sys.executable, __file__, "x"], env=env, capture_output=True, text=True)
        print(out.stdout, out.stderr[-300:], flush=True)
