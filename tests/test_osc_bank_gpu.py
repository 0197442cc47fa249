"""GPU parity: oscillator bank (pwm_update + hard sync of mod_pdm.c:159-175; osc ISR
of mod_osc.c:47-74 with pmeas.h:64-100) against the CPU oracle, bit-exact.
pwm_update / OSC_HARD_SYNC are pinned by the reference itself (mod_pdm.c:159-175 compiled verbatim into
oracle/_ref/libref_pwmosc.so, outputs committed as tests/golden/pwmosc_reference.npz); pmeas_update by
pmeas.h compiled the same way; the sub-oscillator toggle and the ISR around them (mod_osc.c: HAL code
that cannot be built here) are a restatement ("parity unpinned")."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
from synth_tools_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 4, 33, 1000, 1025])
def test_pwmosc_parity_with_hard_sync(smx, orc, n):
    r = synthetic.splitmix64(0x5EED0900 + n, 2 * n).reshape(2, n)
    phase = (r[0] & np.uint64(0xFFFFFF)).astype(np.uint32)
    speed = (np.uint64(256) + r[1] % np.uint64(60000)).astype(np.uint32)
    bank = smx.OscBank(n)
    bank.load_pwm(phase, speed)
    op = phase.copy()
    words = (n + 31) // 32
    rng = np.random.default_rng(n)
    for nt in (1, 63, 200):
        sync = (rng.random((nt, words * 32)) < 0.02)
        sync[:, n:] = False
        bits = np.packbits(sync.reshape(nt, words, 32), axis=2, bitorder="little").view(np.uint32).reshape(nt, words)
        for sb in (None, bits):
            got = bank.tick_n(nt, sb)
            want = np.zeros((nt, n), np.uint8)
            orc.orc_pwmosc_run(op, speed, n, None if sb is None else np.ascontiguousarray(sb).ctypes.data, nt, want.ctypes.data)
            assert np.array_equal(got, want), "n=%d nt=%d" % (n, nt)
    gph, gsp = bank.read_pwm()
    assert np.array_equal(gph, op) and np.array_equal(gsp, speed)
    bank.close()


def test_pwmosc_kernel_reproduces_reference_outputs(smx):
    """Row a-8b on the GPU against the REFERENCE's outputs (no oracle in between): every case of
    tests/golden/pwmosc_reference.npz is one oscillator of a bank, 70 000 ticks, hard syncs as a bit matrix."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pwmosc_reference.npz"))
    names = sorted(k[:-5] for k in g.files if k.endswith("_duty"))
    nt, n = int(g["nticks"]), len(names)
    assert n >= 6
    bank = smx.OscBank(n)
    ph0, sp0 = bank.read_pwm()
    assert ph0.tolist() == [int(g["default_phase"])] * n and sp0.tolist() == [int(g["default_speed"])] * n
    bank.load_pwm(np.array([g[k + "_phase0"] for k in names], np.uint32),
                  np.array([g[k + "_speed"] for k in names], np.uint32))
    sync = np.zeros((nt, 1), np.uint32)
    for c, k in enumerate(names):
        sync[g[k + "_sync_ticks"], 0] |= np.uint32(1 << c)
    # in three calls of ragged length, reading the phases back at a tick the fixture holds
    cuts = [0, 16 * 1000, 16 * 1000 + 16 * 2001, nt]
    for a, b in zip(cuts[:-1], cuts[1:]):
        duty = bank.tick_n(b - a, np.ascontiguousarray(sync[a:b]))
        for c, k in enumerate(names):
            assert np.array_equal(duty[:, c], g[k + "_duty"][a:b]), (k, a)
        ph = bank.read_pwm()[0]
        for c, k in enumerate(names):
            want = int(g[k + "_phase_end"]) if b == nt else int(g[k + "_phase_every16"][b // 16 - 1])
            assert int(ph[c]) == want, (k, b)
    bank.close()


def test_mod_pdm_module_isr(smx, orc):
    """mod_pdm.c's timer ISR as one call (mod_pdm.c:177-194): the carry-out channels, the PWM channel and the
    control trigger every CONTROL_DIV ticks -- against the oracle's two loops (PDM: mod_pdm.c:214-286; pwm_update:
    pinned by the reference itself) and the reference's own CONTROL_DIV (from the compiled :164, in the fixture)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pwmosc_reference.npz"))
    div = int(g["control_div"])
    assert div == 256
    nch, nosc = 2, 1                                       # the firmware's configuration (mod_pdm.c:124-125)
    for nch, nosc in ((2, 1), (70, 33)):
        m = smx.ModPdm(nch, nosc)
        sp, ac = m.pdm.read()
        assert sp[0] == 2000000000 and (sp[1:] == 0x40000000).all() and not ac.any()      # pdm_init, mod_pdm.c:320-326
        ph, spd = m.osc.read_pwm()
        assert not ph.any() and (spd == int(g["default_speed"])).all()
        rng = np.random.default_rng(177194 + nch)
        if nch > 2:
            sp = (0x40000000 + rng.integers(0, 0x80000001, nch)).astype(np.uint32)
            m.pdm.load(sp, ac)
            spd = rng.integers(1, 70000, nosc).astype(np.uint32)
            m.osc.load_pwm(ph, spd)
        o_ac, o_ph = ac.copy(), ph.copy()
        count = triggers = isr = beat = 0
        for nt in (1, 255, 1, 700, 3000):
            dith = (rng.integers(0, 1 << 28, nt)).astype(np.uint32)                        # & 0x0FFFFFFF, mod_pdm.c:261
            sync = rng.random((nt, m.osc.words * 32)) < 0.01
            sync[:, nosc:] = False
            sb = np.ascontiguousarray(np.packbits(sync.reshape(nt, m.osc.words, 32), axis=2, bitorder="little")
                                      .view(np.uint32).reshape(nt, m.osc.words))
            bits, duty, trig = m.tick_n(nt, dith, sb)
            assert np.array_equal(bits, oracle.pdm_run(orc, sp, o_ac, nt, dith))
            want = np.zeros((nt, nosc), np.uint8)
            orc.orc_pwmosc_run(o_ph, spd, nosc, sb.ctypes.data, nt, want.ctypes.data)
            assert np.array_equal(duty, want)
            k = sum(1 for t in range(nt) if (count + t) % div == 0)                       # mod_pdm.c:184-192
            assert trig == k
            for _ in range(k):                                                            # mod_controlrate.c:52-55
                beat += isr % 1024 == 0
                isr += 1
            count = (count + nt) % div
            triggers += k
            assert m.control_div_count == count and m.controlrate()[:2] == (isr, beat)
        assert np.array_equal(m.pdm.read()[1], o_ac) and np.array_equal(m.osc.read_pwm()[0], o_ph)
        assert triggers == (1 + 255 + 1 + 700 + 3000 + div - 1) // div
        m.close()


def test_pwmosc_defaults(smx, orc):
    """pwm_phase 0, pwm_speed 256*13 (mod_pdm.c:160-161)."""
    bank = smx.OscBank(2)
    ph, sp = bank.read_pwm()
    assert ph.tolist() == [0, 0] and sp.tolist() == [3328, 3328]
    duty = bank.tick_n(5000)
    p = np.zeros(1, np.uint32)
    want = [orc.orc_pwm_update(p, 3328) & 0xFF for _ in range(5000)]
    assert duty[:, 0].tolist() == want and duty[:, 1].tolist() == want
    bank.close()


@pytest.mark.parametrize("n", [1, 31, 700])
def test_osc_events_parity(smx, orc, n):
    """Each oscillator gets its own pitch (period in 72 MHz cycles) with jitter; some event
    slots are skipped per oscillator (valid mask); log_max small enough that many averages
    are published, including the u32 division and the double buffer."""
    log_max = 16
    bank = smx.OscBank(n)
    assert bank.set_log_max(0) == -1 and bank.set_log_max(log_max) == 0
    ps = (oracle.Pmeas * n)()
    for c in range(n):
        ps[c].log_max = log_max
    rng = np.random.default_rng(100 + n)
    period = rng.integers(200, 70000, n)
    now = rng.integers(0, 2**32, n, dtype=np.uint64)
    words = (n + 31) // 32
    for call in range(4):
        ne = 37
        valid = rng.random((ne, words * 32)) < 0.8
        valid[:, n:] = False
        cc = np.zeros((ne, n), np.uint32)
        for e in range(ne):
            now = now + np.where(valid[e, :n], period + rng.integers(0, 50, n), 0).astype(np.uint64)
            cc[e] = (now & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        vb = np.packbits(valid.reshape(ne, words, 32), axis=2, bitorder="little").view(np.uint32).reshape(ne, words)
        use_mask = call != 1
        if not use_mask:
            # all-valid call: timestamps must then advance for every oscillator
            cc = (cc.astype(np.uint64) + np.arange(ne, dtype=np.uint64)[:, None] * np.uint64(3)).astype(np.uint32)
        bank.events(cc, vb if use_mask else None)
        orc.orc_osc_bank_events(ps, n, np.ascontiguousarray(cc).reshape(-1),
                                np.ascontiguousarray(vb).ctypes.data if use_mask else None, ne)
        now = cc[-1].astype(np.uint64) | (now & ~np.uint64(0xFFFFFFFF))
    got = bank.read_pmeas()
    want = {"write": [p.write for p in ps], "avg0": [p.avg[0] for p in ps], "avg1": [p.avg[1] for p in ps],
            "num0": [p.num_pub[0] for p in ps], "num1": [p.num_pub[1] for p in ps],
            "num": [p.num for p in ps], "accu": [p.accu for p in ps], "last_cc": [p.last_cc for p in ps],
            "sub": [p.sub for p in ps]}
    for k, v in want.items():
        assert got[k].tolist() == v, k
    assert max(want["write"]) >= 2          # averages were published
    bank.close()


@pytest.mark.parametrize("n", [1, 33, 1500])
def test_clock_bank_parity(smx, orc, n):
    """linux/clock.c:106-120 square-wave / MIDI-clock dividers, incl. hperiod 0 (toggles every
    frame) and the reference's operating point (120 bpm @ 48 kHz -> hperiod 500)."""
    assert smx.lib().smx_bpm_to_hperiod(48000, 120) == orc.orc_bpm_to_hperiod(48000, 120) == 500
    rng = np.random.default_rng(n)
    hp = rng.integers(0, 300, n).astype(np.uint32)
    hp[0] = 500
    if n > 2:
        hp[1], hp[2] = 0, 1
    bank = smx.ClockBank(n)
    _, ph0, po0 = bank.read()
    assert np.all(ph0 == 0) and np.all(po0 == 1)            # clock.c:61-62
    bank.load(hperiod=hp)
    ph, po = np.zeros(n, np.int32), np.ones(n, np.uint32)
    words = (n + 31) // 32
    for nf in (1, 63, 64, 65, 700):
        gp, gt = bank.run(nf)
        wp, wt = np.zeros(nf * words, np.uint32), np.zeros(nf * words, np.uint32)
        orc.orc_clock_run(hp, ph, po, n, nf, wp, wt)
        assert np.array_equal(gp.reshape(-1), wp) and np.array_equal(gt.reshape(-1), wt)
    ghp, gph, gpo = bank.read()
    assert np.array_equal(gph, ph) and np.array_equal(gpo, po) and np.array_equal(ghp, hp)
    bank.close()


def test_clock_bank_loaded_state(smx, orc):
    """Loaded phases beyond the half period (several rolls in a row), polarity words other than
    0/1 (only "zero / non-zero" and "== 1" matter: linux/clock.c:110-116 as restated), waves with and
    without an hperiod of 0, tiles of exactly 64 frames and ragged ones."""
    n = 4096
    rng = np.random.default_rng(9)
    hp = rng.integers(1, 50, n).astype(np.uint32)
    hp[64:128] = rng.integers(0, 3, 64)                        # one wave with zeros
    hp[1000] = 0xFFFFFFFF
    ph = rng.integers(0, 400, n).astype(np.int32)
    ph[5] = -7                                                 # unsigned compare: rolls at once
    po = rng.integers(0, 2, n).astype(np.uint32)
    po[256:320] = rng.integers(0, 6, 64)                       # one wave with other polarity words
    bank = smx.ClockBank(n)
    bank.load(hperiod=hp, phase=ph, pol=po)
    words = (n + 31) // 32
    for nf in (64, 128, 3, 64, 191):
        gp, gt = bank.run(nf)
        wp, wt = np.zeros(nf * words, np.uint32), np.zeros(nf * words, np.uint32)
        orc.orc_clock_run(hp, ph, po, n, nf, wp, wt)
        assert np.array_equal(gp.reshape(-1), wp), nf
        assert np.array_equal(gt.reshape(-1), wt), nf
    _, gph, gpo = bank.read()
    assert np.array_equal(gph, ph) and np.array_equal(gpo, po)
    bank.close()
