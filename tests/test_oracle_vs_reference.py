"""CPU: the oracle of the HEADLINE path against the reference itself.

linux/synth.c:27-208 (note tables, allocator, sum_tick_saw/square, synth_run),
stm32f103/pmeas.h:64-108 (pmeas_update) and stm32f103/mod_pdm.c:159-175 (OSC_HARD_SYNC, pwm_update)
compile verbatim with system headers alone
(oracle/Makefile `ref` streams those line ranges into gcc; nothing is stubbed).  Their
outputs are committed as tests/golden/synth_c_reference.npz / pmeas_reference.npz / pwmosc_reference.npz
(generator: tests/golden/make_golden.py) and the oracle must reproduce them bit for bit;
when oracle/_ref is present (build container, and the GPU box: the .so travels) the oracle
is also compared live on random operation sequences.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle
import replay

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "synth_c_reference.npz"))


def test_note_to_inc_and_midi_tab_equal_reference_outputs(orc, gold):
    assert [orc.orc_note_to_inc(n) for n in range(128)] == gold["note_to_inc"].tolist()
    assert [orc.orc_midi_tab(n) for n in range(128)] == gold["midi_tab"].tolist()
    # note & 127 (linux/synth.c:119)
    assert orc.orc_note_to_inc(128 + 69) == int(gold["note_to_inc"][69])


def test_survey_known_answers_agree_with_the_compiled_reference(gold):
    """The numbers typed in from SURVEY Appendix A.2 are consistent with the real outputs."""
    with open(os.path.join(GOLD, "survey_known_answers.json")) as f:
        kat = json.load(f)
    inc = gold["note_to_inc"]
    assert inc[116:128].tolist() == kat["note_tab"]
    assert all(int(inc[int(n)]) == v for n, v in kat["note_to_inc"].items())
    assert int(inc.astype(np.uint64).sum()) == kat["note_to_inc_sum_0_127"]


@pytest.mark.parametrize("name", replay.SCRIPTS)
def test_oracle_reproduces_reference_scripts(orc, gold, name):
    vec, n2v, inc, st = replay.on_oracle(orc, gold[name + "_script"])
    assert np.array_equal(vec, gold[name + "_vec_bits"])
    assert np.array_equal(n2v, gold[name + "_note2voice"])
    assert np.array_equal(inc, gold[name + "_inc"])
    assert np.array_equal(st, gold[name + "_state"])


def test_fixture_covers_the_quirks(gold):
    """The committed vectors really contain the cases they are meant to pin."""
    s = gold["wrapping_mix_script"]
    assert (s[:, 0] == replay.OP_POKE).sum() >= 3 * 64
    q = gold["b64_quirks_script"]
    assert (q[:, 0] == replay.OP_ON).sum() > 64                   # more notes than voices: steal
    assert gold["b4096_random_vec_bits"].size == 3 * 4096
    assert gold["b1_ticks_script"][:, 1].max() >= 128            # note % 128
    sq = gold["square_vec_bits"].view(np.float32)
    assert set(np.unique(sq[:664]).tolist()) <= {0.0, 0.5}       # sum_tick_square: 0 or 0.5


def test_oracle_pmeas_reproduces_reference_traces(orc):
    g = np.load(os.path.join(GOLD, "pmeas_reference.npz"))
    fields = [str(x) for x in g["fields"]]
    names = sorted(k[:-6] for k in g.files if k.endswith("_trace"))
    assert len(names) >= 6
    published = 0
    for name in names:
        p = oracle.Pmeas(log_max=int(g[name + "_log_max"]))
        for cc, want in zip(g[name + "_cc"], g[name + "_trace"]):
            orc.orc_pmeas_update(C.byref(p), int(cc))
            got = dict(log_max=p.log_max, write=p.write, read=p.read, avg0=p.avg[0], num0=p.num_pub[0],
                       avg1=p.avg[1], num1=p.num_pub[1], num=p.num, accu=p.accu, last_cc=p.last_cc)
            assert [got[f] for f in fields] == want.tolist(), name
        published += p.write
    assert published > 20                                          # the publish branch (pmeas.h:85-91) ran


def pwmosc_cases(g):
    return sorted(k[:-5] for k in g.files if k.endswith("_duty"))


def test_oracle_pwm_update_reproduces_reference_outputs(orc):
    """Row a-8b: orc_pwm_update / orc_pwmosc_run == the reference's pwm_update + OSC_HARD_SYNC
    (mod_pdm.c:159-175 compiled verbatim), 70 000 ticks per case."""
    g = np.load(os.path.join(GOLD, "pwmosc_reference.npz"))
    assert int(g["default_speed"]) == 256 * 13 and int(g["default_phase"]) == 0       # mod_pdm.c:160-161
    assert int(g["control_div"]) == 256                                              # mod_pdm.c:164
    nt = int(g["nticks"])
    names = pwmosc_cases(g)
    assert len(names) >= 6 and nt >= 70000
    for name in names:
        phase = np.array([g[name + "_phase0"]], np.uint32)
        speed = np.array([g[name + "_speed"]], np.uint32)
        sync = np.zeros((nt, 1), np.uint32)
        sync[g[name + "_sync_ticks"], 0] = 1
        # tick by tick through orc_pwm_update for the phases ...
        ph = int(phase[0])
        p1 = np.zeros(1, np.uint32)
        seen = []
        duty1 = np.zeros(nt, np.uint8)
        for t in range(nt):
            if sync[t, 0]:
                ph = 0
            p1[0] = ph
            duty1[t] = orc.orc_pwm_update(p1, int(speed[0])) & 0xFF
            ph = int(p1[0])
            if t % 16 == 15:
                seen.append(ph)
        assert np.array_equal(duty1, g[name + "_duty"]), name
        assert seen == g[name + "_phase_every16"].tolist(), name
        assert ph == int(g[name + "_phase_end"]), name
        # ... and the bank loop the GPU tests use as their oracle
        duty = np.zeros((nt, 1), np.uint8)
        orc.orc_pwmosc_run(phase, speed, 1, sync.ctypes.data, nt, duty.ctypes.data)
        assert np.array_equal(duty[:, 0], g[name + "_duty"]), name
        assert int(phase[0]) == int(g[name + "_phase_end"]), name
        assert int(g[name + "_wraps"]) >= 19, name          # the 24-bit mask (mod_pdm.c:162) is crossed many times


# ---- live, when the compiled reference is present ----------------------------------------
def test_live_random_pwm_updates_against_compiled_mod_pdm_c(orc):
    ref = oracle.load_ref_pwmosc()
    if ref is None:
        pytest.skip("oracle/_ref/libref_pwmosc.so not built (needs /root/reference)")
    rng = np.random.default_rng(159175)
    for case in range(40):
        nt = 5000
        phase0 = int(rng.integers(0, 1 << 24))
        speed = int(rng.integers(0, 1 << int(rng.integers(1, 33))))
        sync = (rng.random(nt) < rng.choice([0.0, 0.001, 0.05])).astype(np.uint8)
        ref.ref_pwm_set(phase0, speed)
        want = np.zeros(nt, np.uint8)
        ref.ref_pwmosc_run(nt, sync.ctypes.data, want.ctypes.data, None)
        phase = np.array([phase0], np.uint32)
        got = np.zeros((nt, 1), np.uint8)
        sync_words = np.ascontiguousarray(sync.astype(np.uint32).reshape(nt, 1))     # kept alive across the call
        orc.orc_pwmosc_run(phase, np.array([speed], np.uint32), 1, sync_words.ctypes.data, nt, got.ctypes.data)
        assert np.array_equal(got[:, 0], want), case
        assert int(phase[0]) == ref.ref_pwm_get_phase(), case


def test_live_random_ops_against_compiled_synth_c(orc):
    ref = oracle.load_ref_synth()
    if ref is None:
        pytest.skip("oracle/_ref/libref_synth.so not built (needs /root/reference)")
    rng = np.random.default_rng(20261004)
    x = oracle.RefSynth()
    ref.synth_init(C.byref(x))
    n2v = np.zeros(128, np.int32)
    inc = np.zeros(64, np.uint32)
    st = np.zeros(64, np.uint32)
    with oracle.quiet_stderr():
        for i in range(3000):
            r = rng.random()
            if r < 0.35:
                n = int(rng.integers(0, 1024))
                ref.synth_note_on(C.byref(x), n)
                orc.orc_note_on(n2v, inc, 64, n)
            elif r < 0.6:
                n = int(rng.integers(0, 1024))
                ref.synth_note_off(C.byref(x), n)
                orc.orc_note_off(n2v, inc, 64, n)
            elif r < 0.65:
                for v in range(64):
                    val = int(rng.integers(0, 2**32))
                    x.voice[v].note_state = val
                    st[v] = val
            elif r < 0.7:
                assert ref.sum_tick_square(C.byref(x)) == orc.orc_sum_tick_square(inc, st, 64)
            elif r < 0.72:
                assert ref.voice_alloc(C.byref(x)) == orc.orc_voice_alloc(inc, 64)
            else:
                nf = int(rng.choice([1, 2, 7, 64, 300]))
                want = np.zeros(nf, np.float32)
                ref.synth_run(C.byref(x), want, nf)
                _, got = oracle.synth_run(orc, inc, st, nf)
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), i
            if i % 100 == 0:
                rn2v, rinc, rst = x.arrays()
                assert np.array_equal(rn2v, n2v) and np.array_equal(rinc, inc) and np.array_equal(rst, st)
    rn2v, rinc, rst = x.arrays()
    assert np.array_equal(rn2v, n2v) and np.array_equal(rinc, inc) and np.array_equal(rst, st)


def test_live_random_timestamps_against_compiled_pmeas_h(orc):
    ref = oracle.load_ref_pmeas()
    if ref is None:
        pytest.skip("oracle/_ref/libref_pmeas.so not built (needs /root/reference)")
    rng = np.random.default_rng(77)
    for lm in (8, 12, 20, 26, 30):
        r = oracle.RefPmeas(ref, lm)
        p = oracle.Pmeas(log_max=lm)
        cc = 0
        for i in range(4000):
            cc = (cc + int(rng.integers(1, 1 << int(rng.integers(1, lm + 2))))) & 0xFFFFFFFF
            r.update(cc)
            orc.orc_pmeas_update(C.byref(p), cc)
            assert (r.get("write"), r.get("num"), r.get("accu"), r.get("last_cc")) == \
                   (p.write, p.num, p.accu, p.last_cc)
        assert [r.get("avg0"), r.get("avg1"), r.get("num0"), r.get("num1")] == \
               [p.avg[0], p.avg[1], p.num_pub[0], p.num_pub[1]]
        assert p.write > 0
