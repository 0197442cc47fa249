"""GPU: model-based fuzz of the PDM, poly, PWM and oscillator banks' C-ABI surface, in the manner of test_saw_api_fuzz_gpu.py.
PDM (stm32f103/mod_pdm.c:198-286): ticks in both output layouts, with and without dither, synchronous and left in
HBM, one- and two-tick launches of big banks (their own kernel), setpoint commands, reloads of either array and
read-backs in random order -- the accumulators are kept lazily (accu0 + T*setpoint + sum of dither), so every call
that reads or replaces them has to settle what the ticks before it left open.  Poly (build-defined):
blocks whose slot fold is left to the next launch, un-fetched blocks, partial reloads and read-backs in between.
SMX_FUZZ_SEED / SMX_FUZZ_ROUNDS widen the run for a soak."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
from synth_tools_amd import synthetic

pytestmark = pytest.mark.gpu


def _to_streams(bits, n):
    nt = bits.shape[0]
    b = ((bits[:, np.arange(n) >> 5] >> (np.arange(n) & 31).astype(np.uint32)) & 1).astype(np.uint32)
    b = b.reshape(nt // 32, 32, n)
    return (b << np.arange(32, dtype=np.uint32)[None, :, None]).sum(axis=1, dtype=np.uint64).astype(np.uint32)


@pytest.mark.parametrize("n", [3, 1500, (1 << 20) + 77])
def test_pdm_every_call_in_random_order(smx, orc, n):
    seed = int(os.environ.get("SMX_FUZZ_SEED", "0xD91"), 0)
    rounds = int(os.environ.get("SMX_FUZZ_ROUNDS", "1"))
    rng = np.random.default_rng(seed + n)
    big = n >= (1 << 20)
    for trial in range(2 * rounds):
        sp, accu = synthetic.pdm_bank(n, 0xD910 + trial + seed)
        accu = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
        bank = smx.PdmBank(n)
        bank.load(sp, accu)
        sp, oa = sp.copy(), accu.copy()
        log = []
        for step in range(30 if big else 60):
            r = rng.random()
            nt = int(rng.choice([1, 1, 2, 2, 3, 8, 31, 64, 65, 100] if not big else [1, 1, 1, 2, 2, 3, 8, 40]))
            d = synthetic.dither_stream(nt, int(rng.integers(1, 1 << 30)), 0x0FFFFFFF) if rng.random() < 0.5 else None
            try:
                if r < 0.45:
                    got = bank.tick_n(nt, d); log.append("tick %d %s" % (nt, "d" if d is not None else "-"))
                    assert np.array_equal(got, oracle.pdm_run(orc, sp, oa, nt, d))
                elif r < 0.55:
                    nt32 = 32 * int(rng.integers(1, 3))
                    d = synthetic.dither_stream(nt32, int(rng.integers(1, 1 << 30)), 0x0FFFFFFF) if d is not None else None
                    got = bank.tick_n_streams(nt32, d); log.append("streams %d" % nt32)
                    assert np.array_equal(got, _to_streams(oracle.pdm_run(orc, sp, oa, nt32, d), n))
                elif r < 0.65:
                    bank.tick_n_async(nt); log.append("async %d" % nt)         # no dither: the device buffer is the caller's
                    oracle.pdm_run(orc, sp, oa, nt, None)
                elif r < 0.70:
                    bank.tick_n(nt, d, want_bits=False); log.append("tick nobits %d" % nt)
                    oracle.pdm_run(orc, sp, oa, nt, d)
                elif r < 0.78:
                    c, v = int(rng.integers(0, n)), int(rng.integers(0, 2**32))
                    assert bank.set_setpoint(c, v) == 0; log.append("setpoint")
                    sp[c] = v
                    assert bank.set_setpoint(n, v) == -2
                elif r < 0.83:
                    sp = rng.integers(0x40000000, 0xC0000000, n, dtype=np.uint64).astype(np.uint32)
                    bank.load(setpoint=sp); log.append("load sp")
                elif r < 0.88:
                    oa = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
                    bank.load(accu=oa); log.append("load accu")
                elif r < 0.94:
                    gsp, gac = bank.read(); log.append("read")
                    assert np.array_equal(gsp, sp) and np.array_equal(gac, oa)
                else:
                    bank.sync(); log.append("sync")
            except AssertionError:
                raise AssertionError("n=%d trial=%d step=%d after: %s" % (n, trial, step, " | ".join(log[-12:])))
        gsp, gac = bank.read()
        assert np.array_equal(gsp, sp) and np.array_equal(gac, oa), " | ".join(log[-12:])
        bank.close()


@pytest.mark.parametrize("n", [700, (1 << 18) + 5])
def test_poly_every_call_in_random_order(smx, orc, inc_table, n):
    seed = int(os.environ.get("SMX_FUZZ_SEED", "0xE91"), 0)
    rounds = int(os.environ.get("SMX_FUZZ_ROUNDS", "1"))
    rng = np.random.default_rng(seed + n)
    for trial in range(2 * rounds):
        arrs = synthetic.poly_bank(n, 0xE910 + trial + seed, inc_table, active_fraction=0.9)
        bank = smx.PolyBank(n)
        bank.load(**arrs)
        keep = {k: v.copy() for k, v in arrs.items()}
        ob = oracle.PolyBank(n=n, **{k: v.ctypes.data for k, v in keep.items()})
        log = []
        for step in range(40):
            r = rng.random()
            nf = int(rng.choice([1, 2, 7, 32, 63, 64, 64, 65, 130]))
            try:
                if r < 0.40:
                    bus, vec = bank.run(nf); log.append("run %d" % nf)
                    want = np.zeros(2 * nf, np.int32)
                    orc.orc_poly_run(C.byref(ob), want, nf)
                    assert np.array_equal(bus.reshape(-1), want)
                elif r < 0.65:
                    nf = min(nf, 64)
                    bank.run_async(nf); log.append("async %d" % nf)           # its fold may be left to the next launch
                    orc.orc_poly_run(C.byref(ob), np.zeros(2 * nf, np.int32), nf)
                elif r < 0.78:
                    flip = rng.random(n) < 0.2                                # control-rate input: gates
                    keep["gate"][:] = np.where(flip, 1 - keep["gate"], keep["gate"])
                    bank.load(gate=keep["gate"]); log.append("gates")
                elif r < 0.84:
                    keep["pan"][:] = rng.integers(0, 1 << 16, n).astype(keep["pan"].dtype)
                    keep["inc"][:] = np.where(rng.random(n) < 0.1, 0, keep["inc"])
                    bank.load(pan=keep["pan"], inc=keep["inc"]); log.append("pan+inc")
                elif r < 0.92:
                    got = bank.read(); log.append("read")
                    for k in ("phase", "level", "stage", "inc", "gate", "pan"):
                        assert np.array_equal(got[k], keep[k]), k
                    assert np.array_equal(got["y"].view(np.uint32), keep["y"].view(np.uint32))
                else:
                    bank.sync(); log.append("sync")
            except AssertionError:
                raise AssertionError("n=%d trial=%d step=%d after: %s" % (n, trial, step, " | ".join(log[-12:])))
        got = bank.read()
        for k in ("phase", "level", "stage", "inc", "gate", "ar", "dr", "sl", "rr", "pan"):
            assert np.array_equal(got[k], keep[k]), k
        assert np.array_equal(got["y"].view(np.uint32), keep["y"].view(np.uint32))
        bank.close()


def _pwm_state(n, order, rng):
    u = lambda: rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    v = lambda: ((rng.integers(0, 1 << 20, n, dtype=np.uint64)).astype(np.uint32) - np.uint32(1 << 19))
    arrs = dict(setpoint=u(), pos0=u(), vel0=v(), pos1=u(), vel1=v())
    for k in range(order):
        arrs["s%d" % (k + 1)] = u()
    return arrs


@pytest.mark.parametrize("n,order", [(5, 2), (3000, 1), (3000, 3), ((1 << 18) + 9, 2), (1500, 4)])
def test_pwm_every_call_in_random_order(smx, orc, n, order):
    """The noise-shaped PWM bank (mod_pdm_pwm.c:80-143, pdm.h, mod_controlrate.c:28-57): ticks across control-rate
    boundaries with and without dither and with the duty bytes kept in HBM, setpoint commands, partial reloads, the
    divider set from outside, read-backs -- against the oracle's bank driven by the same calls."""
    seed = int(os.environ.get("SMX_FUZZ_SEED", "0xF91"), 0)
    rng = np.random.default_rng(seed + 7 * n + order)
    for trial in range(2 * int(os.environ.get("SMX_FUZZ_ROUNDS", "1"))):
        div_log, sh = int(rng.integers(3, 8)), int(rng.choice([24, 24, 16]))
        arrs = _pwm_state(n, order, rng)
        bank = smx.PwmBank(n, order=order, control_div_log=div_log, out_shift=sh)
        bank.load(**arrs)
        keep = {k: v.copy() for k, v in arrs.items()}
        for k in ("s1", "s2", "s3", "s4"):
            keep.setdefault(k, np.zeros(n, np.uint32))
        ob = oracle.PwmBank(n=n, order=order, div_count=0, div_log=div_log, out_shift=sh,
                            s=(C.c_void_p * 4)(*[keep["s%d" % k].ctypes.data for k in (1, 2, 3, 4)]),
                            **{k: keep[k].ctypes.data for k in ("setpoint", "pos0", "vel0", "pos1", "vel1")})
        log = []
        for step in range(40):
            r = rng.random()
            nt = int(rng.choice([1, 2, 7, 31, 32, 33, 100, 257]))
            d = synthetic.dither_stream(nt, int(rng.integers(1, 1 << 30)), 0x3FF) if rng.random() < 0.6 else None
            try:
                if r < 0.45:
                    got = bank.tick_n(nt, d); log.append("tick %d" % nt)
                    want = np.zeros((nt, n), np.uint8)
                    orc.orc_pwm_bank_run(C.byref(ob), None if d is None else d.ctypes.data, nt, want.ctypes.data)
                    assert np.array_equal(got, want)
                elif r < 0.60:
                    bank.tick_n(nt, d, want_duty=False); log.append("tick nobytes %d" % nt)
                    orc.orc_pwm_bank_run(C.byref(ob), None if d is None else d.ctypes.data, nt, None)
                elif r < 0.70:
                    bank.tick_n_async(nt); log.append("async %d" % nt)
                    orc.orc_pwm_bank_run(C.byref(ob), None, nt, None)
                elif r < 0.78:
                    c, v = int(rng.integers(0, n)), int(rng.integers(0, 2**32))
                    assert bank.set_setpoint(c, v) == 0; log.append("setpoint")
                    keep["setpoint"][c] = v
                    assert bank.set_setpoint(n, v) == -2
                elif r < 0.84:
                    keep["setpoint"][:] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
                    keep["vel1"][:] = (rng.integers(0, 1 << 20, n, dtype=np.uint64)).astype(np.uint32) - np.uint32(1 << 19)
                    bank.load(setpoint=keep["setpoint"], vel1=keep["vel1"]); log.append("load")
                elif r < 0.88:
                    c = int(rng.integers(0, 1 << div_log))
                    bank.div_count = c; ob.div_count = c; log.append("div %d" % c)
                elif r < 0.95:
                    st = bank.read(); log.append("read")
                    for k in st:
                        assert np.array_equal(st[k], keep[k].view(np.uint32)), k
                else:
                    bank.sync(); log.append("sync")
                assert bank.div_count == ob.div_count
            except AssertionError:
                raise AssertionError("n=%d order=%d trial=%d step=%d after: %s" % (n, order, trial, step, " | ".join(log[-12:])))
        st = bank.read()
        for k in st:
            assert np.array_equal(st[k], keep[k].view(np.uint32)), (k, " | ".join(log[-12:]))
        bank.close()


@pytest.mark.parametrize("n", [3, 1000, (1 << 16) + 5])
def test_pwmosc_every_call_in_random_order(smx, orc, n):
    """The PWM oscillators (mod_pdm.c:159-175; pwm_update pinned by the compiled reference): ticks with and without a
    hard-sync matrix, duty kept or dropped, phases / speeds reloaded one at a time, read-backs."""
    seed = int(os.environ.get("SMX_FUZZ_SEED", "0xC91"), 0)
    rng = np.random.default_rng(seed + n)
    words = (n + 31) // 32
    for trial in range(2 * int(os.environ.get("SMX_FUZZ_ROUNDS", "1"))):
        phase = rng.integers(0, 1 << 24, n).astype(np.uint32)
        speed = rng.integers(1, 70000, n).astype(np.uint32)
        bank = smx.OscBank(n)
        bank.load_pwm(phase, speed)
        op = phase.copy()
        log = []
        for step in range(40):
            r = rng.random()
            nt = int(rng.choice([1, 2, 8, 9, 63, 64, 200]))
            sb = None
            if rng.random() < 0.5:
                sync = rng.random((nt, words * 32)) < 0.03
                sync[:, n:] = False
                sb = np.ascontiguousarray(np.packbits(sync.reshape(nt, words, 32), axis=2, bitorder="little")
                                          .view(np.uint32).reshape(nt, words))
            try:
                if r < 0.6:
                    want_duty = rng.random() < 0.7
                    got = bank.tick_n(nt, sb, want_duty=want_duty); log.append("tick %d" % nt)
                    want = np.zeros((nt, n), np.uint8)
                    orc.orc_pwmosc_run(op, speed, n, None if sb is None else sb.ctypes.data, nt, want.ctypes.data)
                    if want_duty:
                        assert np.array_equal(got, want)
                elif r < 0.72:
                    speed = rng.integers(1, 70000, n).astype(np.uint32)
                    bank.load_pwm(speed=speed); log.append("load speed")
                elif r < 0.84:
                    op = rng.integers(0, 1 << 24, n).astype(np.uint32)
                    bank.load_pwm(phase=op); log.append("load phase")
                else:
                    gph, gsp = bank.read_pwm(); log.append("read")
                    assert np.array_equal(gph, op) and np.array_equal(gsp, speed)
            except AssertionError:
                raise AssertionError("n=%d trial=%d step=%d after: %s" % (n, trial, step, " | ".join(log[-12:])))
        gph, gsp = bank.read_pwm()
        assert np.array_equal(gph, op) and np.array_equal(gsp, speed)
        bank.close()
