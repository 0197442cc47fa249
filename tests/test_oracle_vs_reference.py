"""CPU: the oracle of the HEADLINE path against the reference itself.

linux/synth.c:27-208 (note tables, allocator, sum_tick_saw/square, synth_run) and
stm32f103/pmeas.h:64-108 (pmeas_update) compile verbatim with system headers alone
(oracle/Makefile `ref` streams those line ranges into gcc; nothing is stubbed).  Their
outputs are committed as tests/golden/synth_c_reference.npz / pmeas_reference.npz
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


# ---- live, when the compiled reference is present ----------------------------------------
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
