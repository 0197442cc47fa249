"""Blocks of 17..32 frames of big banks: the direct form (form STEPPING pinned by the caller = what every block of up to
32 frames took before), AUTO (one 32-frame chunk: the device-side flag picks stepping or located wraps) and the event
form pinned -- checked against the closed form of the linear phasor, then timed in a stream of un-fetched blocks.
    python tools/explore_short_events.py            -> us per block, per bank / frame count / form
    SMX_SAW_CARRY_MIN_LOG2=28 python tools/...      -> the same with smaller banks admitted to the chunk path
    LGS=26 FRAMES=128,192,256 python tools/...       -> other banks (log2 of the voice count) / block lengths"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth_tools_amd as sta
from synth_tools_amd import synthetic


def bus_at(inc, st, frames):
    on = inc != 0
    out = []
    with np.errstate(over="ignore"):
        for f in frames:
            ph = st + np.uint32(f) * inc
            out.append(int(np.where(on, ph.view(np.int32) >> 4, 0).sum(dtype=np.int64)))
    return ((np.array(out, np.int64) + (1 << 31)) % (1 << 32) - (1 << 31)).astype(np.int32)


def main():
    tab = synthetic.note_inc_table(sta.lib().note_to_inc)
    lgs = [int(x) for x in os.environ.get("LGS", "25,26").split(",")]
    for lg in lgs:
        n = 1 << lg
        for kind in ("piano", "high"):
            inc, st = synthetic.saw_bank(n, 0x5EED0005, tab)
            if kind == "high":
                inc = (inc | np.uint32(0xC0000000)).astype(np.uint32)         # every voice above the event form's bound
            b = sta.SawBank(n)
            b.load(inc, st)
            for nf in [int(x) for x in os.environ.get("FRAMES", "17,24,32,64").split(",")]:
                row = {}
                for form, name in ((1, "direct"), (0, "auto"), (2, "events")):
                    if kind == "high" and form == 2 and nf >= 64:
                        continue
                    b.set_block_form(form)
                    b.load(state=st)
                    for _ in range(6):                                          # AUTO: measure, then pin
                        b.run_async(nf)
                    b.sync()
                    b.load(state=st)
                    b.run_async(nf)
                    got = b.fetch(nf)[0]
                    pick = [0, nf // 2, nf - 1]
                    ok = bool(np.array_equal(got[pick], bus_at(inc, st, pick)))
                    for _ in range(30): b.run_async(nf)
                    b.sync()
                    best = 1e9
                    for rep in range(5):
                        b.timer_start()
                        for _ in range(40): b.run_async(nf)
                        best = min(best, b.timer_stop() / 40)
                    row[name] = (round(best * 1e3, 1), "ok" if ok else "WRONG")
                print("2^%d %-5s x %2d frames: %s   hbm %%: %s" % (
                    lg, kind, nf, json.dumps(row),
                    {k: round(100 * 8 * n / (v[0] * 1e-6) / 8e12, 1) for k, v in row.items()}), flush=True)
            b.close()


if __name__ == "__main__":
    main()
