"""GPU parity: noise-shaped PWM bank (mod_pdm_pwm.c ISR + pdm.h shapers +
mod_controlrate.c line update) against the CPU oracle, bit-exact.

The shapers pdm1..4 in the oracle are pinned against the reference's real pdm.h
(tests/test_oracle_golden.py); the ISR/control composition is a restatement of ARM/HAL
code that cannot be built here ("parity unpinned" for the composition)."""
import ctypes as C

import numpy as np
import pytest

import oracle
from synth_tools_amd import synthetic

pytestmark = pytest.mark.gpu


def _oracle_bank(n, arrs, order, div_log, out_shift, div_count=0):
    keep = {k: v.copy() for k, v in arrs.items()}
    for k in ("s1", "s2", "s3", "s4"):
        keep.setdefault(k, np.zeros(n, np.uint32))
    b = oracle.PwmBank(n=n, order=order, div_count=div_count, div_log=div_log, out_shift=out_shift,
                       s=(C.c_void_p * 4)(*[keep["s%d" % k].ctypes.data for k in (1, 2, 3, 4)]),
                       **{k: keep[k].ctypes.data for k in ("setpoint", "pos0", "vel0", "pos1", "vel1")})
    return b, keep


def _random_state(n, order, seed):
    r = synthetic.splitmix64(seed, 9 * n).reshape(9, n)
    u = lambda k: (r[k] >> np.uint64(32)).astype(np.uint32)
    arrs = dict(setpoint=u(0), pos0=u(1), vel0=(u(2) >> np.uint32(12)) - np.uint32(1 << 19),
                pos1=u(3), vel1=(u(4) >> np.uint32(12)) - np.uint32(1 << 19))
    for k in range(order):
        arrs["s%d" % (k + 1)] = u(5 + k)
    return arrs


@pytest.mark.parametrize("order", [1, 2, 3, 4])
@pytest.mark.parametrize("n", [1, 3, 5, 1023, 1025, 4100])
def test_parity_orders_and_ragged(smx, orc, n, order):
    div_log, sh = 5, 24
    arrs = _random_state(n, order, 0x5EED0800 + 16 * n + order)
    bank = smx.PwmBank(n, order=order, control_div_log=div_log, out_shift=sh)
    bank.load(**arrs)
    ob, keep = _oracle_bank(n, arrs, order, div_log, sh)
    for k, nt in enumerate([1, 31, 32, 33, 100, 7]):           # crosses control-rate boundaries mid-call
        d = synthetic.dither_stream(nt, 50 + k, 0x3FF) if k % 2 else None     # mod_pdm_pwm.c:127
        got = bank.tick_n(nt, d)
        want = np.zeros((nt, n), np.uint8)
        orc.orc_pwm_bank_run(C.byref(ob), None if d is None else d.ctypes.data, nt, want.ctypes.data)
        assert np.array_equal(got, want), "n=%d order=%d nt=%d" % (n, order, nt)
        assert bank.div_count == ob.div_count
    st = bank.read()
    for k in st:
        assert np.array_equal(st[k], keep[k].view(np.uint32)), k
    bank.close()


def test_firmware_config(smx, orc):
    """The firmware's own configuration: 3 channels (mod_pdm_pwm.c:42-43), order 2,
    CONTROL_DIV_LOG 12, out_shift 24, pdm_init setpoints, dither masked 0x3FF; run across
    two control periods with a SETPOINT command in between (mod_synth.c:104-111)."""
    bank = smx.PwmBank(3)
    bank.init()
    st = bank.read()
    assert st["setpoint"].tolist() == [2000000000, 0x40000000, 0x40000000]
    ob, keep = _oracle_bank(3, {k: st[k] for k in ("setpoint", "pos0", "vel0", "pos1", "vel1")}, 2, 12, 24)
    for blk in range(3):
        nt = 4096 + 17
        d = synthetic.dither_stream(nt, 0xD17 + blk, 0x3FF)
        got = bank.tick_n(nt, d)
        want = np.zeros((nt, 3), np.uint8)
        orc.orc_pwm_bank_run(C.byref(ob), d.ctypes.data, nt, want.ctypes.data)
        assert np.array_equal(got, want)
        if blk == 0:
            assert bank.set_setpoint(3, 1) == -2
            assert bank.set_setpoint(1, 0xA0000000) == 0
            keep["setpoint"][1] = 0xA0000000
    st = bank.read()
    for k in st:
        assert np.array_equal(st[k], keep[k].view(np.uint32)), k
    # the glide has moved channel 0 towards its setpoint and the duty tracks position >> 24
    assert abs(int(got[-64:, 0].astype(np.int64).mean()) - (int(keep["pos0"][0]) >> 24)) <= 1
    bank.close()


def test_full_size_properties_256k_channels(smx, orc):
    """256 Ki channels x 1024 ticks: a slice against the oracle and, for every channel,
    the noise-shaper invariant of pdm.h:13-40 with dither 0: the integrator chain telescopes, so
    s1' - s1 = sum(pos0_t) - (sum(q_t) << sh)  (mod 2^32)."""
    n, nt, sh = 1 << 18, 1024, 24
    arrs = _random_state(n, 2, 0x5EED0808)
    arrs["vel0"][:] = 0
    arrs["vel1"][:] = 0
    arrs["pos1"][:] = arrs["pos0"]
    arrs["setpoint"][:] = arrs["pos0"]       # static lines: pos0 constant over the run
    bank = smx.PwmBank(n, order=2, control_div_log=12, out_shift=sh)
    bank.load(**arrs)
    duty = bank.tick_n(nt)
    st = bank.read()
    qsum = duty.astype(np.uint64).sum(axis=0)
    lhs = (st["s1"] - arrs["s1"]).astype(np.uint32)
    rhs = ((arrs["pos0"].astype(np.uint64) * np.uint64(nt) - (qsum << np.uint64(sh))) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    assert np.array_equal(lhs, rhs)
    sl = slice(7 * 1024, 7 * 1024 + 512)
    ob, keep = _oracle_bank(512, {k: np.ascontiguousarray(v[sl]) for k, v in arrs.items()}, 2, 12, sh)
    want = np.zeros((nt, 512), np.uint8)
    orc.orc_pwm_bank_run(C.byref(ob), None, nt, want.ctypes.data)
    assert np.array_equal(duty[:, sl], want)
    bank.close()


def test_controlrate_beat_divider(smx):
    """control_update's beat divider (mod_controlrate.c:19, 52-55): `if (isr_count % 1024 == 0) beat_pulse++;
    isr_count++` once per control tick, a control tick being a sample tick that starts with
    control_div_count == 0 (mod_pdm_pwm.c:129-137); controlrate_poll handles one beat per call (:64-72).
    Ragged runs over more than 2 x 1024 control ticks against the statement above."""
    div_log = 2
    bank = smx.PwmBank(96, order=2, control_div_log=div_log)
    bank.init()
    assert bank.controlrate() == (0, 0, 0)
    rng = np.random.default_rng(5)
    div_count, isr, beat = 0, 0, 0
    handled = 0
    total = 0
    while isr < 2 * 1024 + 300:
        n = int(rng.choice([1, 2, 3, 4, 5, 64, 257, 1000]))
        for _ in range(n):
            if div_count == 0:
                if isr % 1024 == 0:
                    beat += 1
                isr += 1
            div_count = (div_count + 1) % (1 << div_log)
        bank.tick_n(n, want_duty=False)
        total += n
        assert bank.div_count == div_count
        got = bank.controlrate()
        assert got[:2] == (isr, beat), (total, got)
        if rng.random() < 0.3 and handled < beat:
            assert bank.controlrate_poll() == 1
            handled += 1
            assert bank.controlrate()[2] == handled
    assert beat == 3
    while handled < beat:
        assert bank.controlrate_poll() == 1
        handled += 1
    assert bank.controlrate_poll() == 0 and bank.controlrate() == (isr, beat, beat)
    bank.init()
    assert bank.controlrate() == (0, 0, 0)
    bank.close()
