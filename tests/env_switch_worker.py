"""One process with one of the library's A/B switches set (test helper for tests/test_env_switches_gpu.py): blocks
that reach the code path the switch selects, compared with the oracle.  Prints "ok <checks>"."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import oracle
    import synth_tools_amd as sta
    from synth_tools_amd import synthetic
    orc = oracle.load()
    tab = synthetic.note_inc_table(sta.lib().note_to_inc)
    checks = 0
    # (voices, block lengths): direct slot launches, tick launches, carry forms (stepping and events), small banks
    for n, frames, form in (((1 << 20) + 3, (8, 16, 64, 64, 3, 100, 32), 0),
                            (1 << 23, (64, 128, 64), 1),               # carry, stepping pinned
                            (1 << 23, (64, 300, 64, 100, 128, 65), 2), # carry, events pinned (incl. a long launch and 65..128 frames)
                            (70000, (64, 1, 17, 200), 0)):
        inc, st = synthetic.saw_bank(n, 0x5EED0E00 + n % 977, tab, active_fraction=0.9)
        bank = sta.SawBank(n)
        bank.load(inc, st)
        bank.set_block_form(form)
        for k, nf in enumerate(frames):
            bank.run_async(nf)
            if n >= (1 << 23):
                # 2^23 voices: three frames of the bus against the closed form of the linear phasor (stepping them on
                # the CPU in fourteen processes is most of this test's time); the final phases are compared below
                pick = sorted({0, nf // 2, nf - 1})
                with np.errstate(over="ignore"):
                    w = [int(np.where(inc != 0, (st + np.uint32(f) * inc).view(np.int32) >> 4, 0).sum(dtype=np.int64)) for f in pick]
                    st += np.uint32(nf) * inc
                want = ((np.array(w, np.int64) + (1 << 31)) % (1 << 32) - (1 << 31)).astype(np.int32)
            else:
                pick = slice(None)
                want, _ = oracle.synth_run(orc, inc, st, nf, want_vec=False)
            if k % 2 == 1 or k == len(frames) - 1:          # un-fetched blocks in between
                bus, _ = bank.fetch(nf)
                assert np.array_equal(bus[pick], want), (n, nf, form)
                checks += 1
        assert np.array_equal(bank.read()[1], st), (n, "phases")
        bank.close()
    # blocks of 17..32 frames of a 2^25-voice bank: one 32-frame chunk of the carry forms (SMX_SAW_NO_SHORT_EVENTS: the
    # direct form); three frames per block against the closed form of the linear phasor
    # (only in the processes whose switch touches that path, and in the one without a switch: the bank takes seconds to make)
    other = ("SMX_POLY_NO_DEFER", "SMX_PDM_NO_FEWTICKS", "SMX_NO_PUBLISH", "SMX_BANK_TWO_ALLOCS", "SMX_SAW_NO_LONG_EVENTS",
             "SMX_SAW_NO_SLOTS", "SMX_SAW_NO_DEFER", "SMX_SAW_NO_EVENTS_128")
    n = 1 << 25
    inc, st = synthetic.saw_bank(n, 0x5EED0E05, tab, active_fraction=0.9) if not any(os.environ.get(k) for k in other) else (None, None)
    bank = sta.SawBank(n if inc is not None else 64)
    if inc is not None:
        bank.load(inc, st)
    for k, nf in enumerate((32, 20, 32, 32) if inc is not None else ()):
        bank.run_async(nf)
        if k != 1:
            pick = [0, nf // 2, nf - 1]
            with np.errstate(over="ignore"):
                want = [int(np.where(inc != 0, (st + np.uint32(f) * inc).view(np.int32) >> 4, 0).sum(dtype=np.int64)) for f in pick]
            want = ((np.array(want, np.int64) + (1 << 31)) % (1 << 32) - (1 << 31)).astype(np.int32)
            assert np.array_equal(bank.fetch(nf)[0][pick], want), ("short chunk", k, nf)
            checks += 1
        with np.errstate(over="ignore"):
            st += np.uint32(nf) * inc
    assert inc is None or np.array_equal(bank.read()[1], st), "short chunk phases"
    bank.close()
    # the drop-in synth_run on a caller-owned struct synth (one launch that publishes its own bus; SMX_NO_PUBLISH:
    # upload + bank kernel + copy), against the REFERENCE's committed outputs: 1-, 64-, 256- and 4096-frame blocks
    import replay
    gold = np.load(os.path.join(ROOT, "tests", "golden", "synth_c_reference.npz"))
    for name in ("b1_ticks", "b64_quirks", "b4096_random", "wrapping_mix"):
        vec, n2v, inc, st = replay.on_struct_synth(sta.lib(), sta.Synth, gold[name + "_script"], square=sta.lib().sum_tick_square)
        assert np.array_equal(vec, gold[name + "_vec_bits"]), name
        assert np.array_equal(st, gold[name + "_state"]) and np.array_equal(inc, gold[name + "_inc"]), name
        checks += 1
    # the PDM bank in the tick regime on a big bank (<= 8 ticks per launch: the few-ticks read-stream kernel;
    # SMX_PDM_NO_FEWTICKS: the tile kernel), with and without dither, the lazily kept accumulators read back in between
    nch = (1 << 20) + 1024 + 37
    sp, ac = synthetic.pdm_bank(nch, 0x5EED0E03)
    ac = (synthetic.splitmix64(9, nch) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    pdm = sta.PdmBank(nch)
    pdm.load(sp, ac)
    oac = ac.copy()
    for k, nt in enumerate((1, 3, 8, 64, 1, 5, 2, 100, 7)):
        dith = synthetic.dither_stream(nt, 40 + k, 0x0FFFFFFF) if k % 2 else None
        bits = pdm.tick_n(nt, dith)
        assert np.array_equal(bits, oracle.pdm_run(orc, sp, oac, nt, dith)), ("pdm", k, nt)
        if k in (2, 5, 8):
            assert np.array_equal(pdm.read()[1], oac), ("pdm accu", k)
        checks += 1
    pdm.close()
    # the poly bank: un-fetched blocks, then a fetched one (the fold deferred to the next launch; SMX_POLY_NO_DEFER:
    # every launch folds its own copies)
    import ctypes as C
    n = 5000
    a = synthetic.poly_bank(n, 0x5EED0E04, tab, active_fraction=0.9)
    pb = sta.PolyBank(n)
    pb.load(**a)
    ob = oracle.PolyBank(n=n, **{k: v.ctypes.data for k, v in a.items()})
    for k, nf in enumerate((64, 64, 7, 64, 1, 64)):
        want = np.zeros(2 * nf, np.int32)
        orc.orc_poly_run(C.byref(ob), want, nf)
        if k in (2, 3, 5):
            got, _ = pb.run(nf)
            assert np.array_equal(got.reshape(-1), want), ("poly", k, nf)
            checks += 1
        else:
            pb.run_async(nf)
    got = pb.read()
    for f in ("phase", "level", "stage"):
        assert np.array_equal(got[f], a[f]), ("poly state", f)
    assert np.array_equal(got["y"].view(np.uint32), a["y"].view(np.uint32))
    pb.close()
    print("ok", checks)


if __name__ == "__main__":
    main()
