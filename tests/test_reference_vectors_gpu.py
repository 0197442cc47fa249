"""GPU parity against the REFERENCE's own outputs (not only against the oracle).

tests/golden/synth_c_reference.npz holds what /root/reference/linux/synth.c:27-208, compiled
verbatim, produced for scripted note-on/off/run sequences; pmeas_reference.npz what
stm32f103/pmeas.h:64-108 produced.  The product's drop-in entry points (same names, same
1024-byte struct synth) and its N-voice bank must give the same bits; with oracle/_ref
present (the prebuilt .so travels to the GPU box) they are also compared live."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
import replay

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "synth_c_reference.npz"))


def test_note_to_inc_table(smx, gold):
    L = smx.lib()
    assert [L.note_to_inc(n) for n in range(128)] == gold["note_to_inc"].tolist()
    tab = (C.c_uint8 * 128).in_dll(L, "midi_tab")
    assert list(tab) == gold["midi_tab"].tolist()


@pytest.mark.parametrize("name", replay.SCRIPTS)
def test_dropin_reproduces_reference_scripts(smx, gold, name):
    """synth_init / synth_note_on / synth_note_off / synth_run / sum_tick_square of
    libsynth_mi355x.so on a caller-owned struct synth == linux/synth.c's outputs."""
    L = smx.lib()
    vec, n2v, inc, st = replay.on_struct_synth(L, smx.Synth, gold[name + "_script"], square=L.sum_tick_square)
    assert np.array_equal(vec, gold[name + "_vec_bits"])
    assert np.array_equal(n2v, gold[name + "_note2voice"])
    assert np.array_equal(inc, gold[name + "_inc"])
    assert np.array_equal(st, gold[name + "_state"])


@pytest.mark.parametrize("name", ["b64_quirks", "b4096_random", "b1_ticks"])
def test_bank_of_64_voices_reproduces_reference_scripts(smx, gold, name):
    """The N-voice bank API at N = 64 (allocator + kernels in HBM) == linux/synth.c."""
    bank = smx.SawBank(64)
    out = []
    for op, a, b in gold[name + "_script"]:
        a = int(a)
        if op == replay.OP_ON:
            bank.note_on(a)
        elif op == replay.OP_OFF:
            bank.note_off(a)
        elif op == replay.OP_RUN:
            out.append(bank.run(a)[1])
    inc, st = bank.read()
    bank.close()
    assert np.array_equal(np.concatenate(out).view(np.uint32), gold[name + "_vec_bits"])
    assert np.array_equal(inc, gold[name + "_inc"]) and np.array_equal(st, gold[name + "_state"])


def test_bank_wrapping_mix_with_loaded_phases(smx, gold):
    """>= 16 full-scale voices: the reference's `int sum` wraps; the bank's integer bus must wrap
    the same way (phases poked through smx_bank_load, as the script pokes voice[].note_state)."""
    bank = smx.SawBank(64)
    out = []
    pokes = {}
    for op, a, b in gold["wrapping_mix_script"]:
        a = int(a)
        if op == replay.OP_ON:
            bank.note_on(a)
        elif op == replay.OP_POKE:
            pokes[a] = int(b)
        elif op == replay.OP_RUN:
            if pokes:
                inc, st = bank.read()
                for v, val in pokes.items():
                    st[v] = val
                bank.load(inc, st)
                pokes = {}
            out.append(bank.run(a)[1])
    inc, st = bank.read()
    bank.close()
    assert np.array_equal(np.concatenate(out).view(np.uint32), gold["wrapping_mix_vec_bits"])
    assert np.array_equal(st, gold["wrapping_mix_state"])


def test_osc_events_reproduce_reference_pmeas_traces(smx):
    """osc_events_kernel (pmeas state machine on the GPU) == pmeas.h:64-108's outputs: every trace of
    the fixture is one oscillator of a bank, all traces run side by side."""
    g = np.load(os.path.join(GOLD, "pmeas_reference.npz"))
    names = sorted(k[:-6] for k in g.files if k.endswith("_trace"))
    fields = [str(x) for x in g["fields"]]
    for name in names:
        lm = int(g[name + "_log_max"])
        cc = g[name + "_cc"]
        want = g[name + "_trace"]
        # replicate the trace over 70 oscillators (more than a wave) and check checkpoints
        n = 70
        bank = smx.OscBank(n)
        assert bank.set_log_max(lm) == 0
        cuts = [0, 1, 2, len(cc) // 3, len(cc) // 3 + 1, len(cc) - 7, len(cc)]
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            if hi <= lo:
                continue
            ev = np.repeat(cc[lo:hi, None], n, axis=1)
            bank.events(np.ascontiguousarray(ev))
            got = bank.read_pmeas()
            w = dict(zip(fields, want[hi - 1].tolist()))
            for c in (0, 63, 64, n - 1):
                assert int(got["write"][c]) == w["write"], (name, hi)
                assert (int(got["avg0"][c]), int(got["num0"][c]), int(got["avg1"][c]), int(got["num1"][c])) == \
                       (w["avg0"], w["num0"], w["avg1"], w["num1"]), (name, hi)
                assert (int(got["num"][c]), int(got["accu"][c]), int(got["last_cc"][c])) == \
                       (w["num"], w["accu"], w["last_cc"]), (name, hi)
        bank.close()


def test_live_dropin_against_compiled_synth_c(smx):
    ref = oracle.load_ref_synth()
    if ref is None:
        pytest.skip("oracle/_ref/libref_synth.so not present")
    L = smx.lib()
    rng = np.random.default_rng(4242)
    xr, xp = oracle.RefSynth(), smx.Synth()
    ref.synth_init(C.byref(xr))
    L.synth_init(C.byref(xp))
    with oracle.quiet_stderr():
        for i in range(400):
            r = rng.random()
            n = int(rng.integers(0, 512))
            if r < 0.4:
                ref.synth_note_on(C.byref(xr), n); L.synth_note_on(C.byref(xp), n)
            elif r < 0.6:
                ref.synth_note_off(C.byref(xr), n); L.synth_note_off(C.byref(xp), n)
            elif r < 0.65:
                for v in range(64):
                    val = int(rng.integers(0, 2**32))
                    xr.voice[v].note_state = val
                    xp.voice[v].note_state = val
            elif r < 0.7:
                assert ref.sum_tick_square(C.byref(xr)) == L.sum_tick_square(C.byref(xp))
            else:
                nf = int(rng.choice([1, 3, 64, 257]))
                a, b = np.zeros(nf, np.float32), np.zeros(nf, np.float32)
                ref.synth_run(C.byref(xr), a, nf)
                L.synth_run(C.byref(xp), b, nf)
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), i
    assert bytes(xr) == bytes(xp)
