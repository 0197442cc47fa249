"""CPU: the closed forms bench.py checks the timed kernels' outputs against (numpy, inside bench.py because the bench
may use oracle/ only for its CPU baseline) agree with the oracle -- so a bench that passes its own check has a bus
that the oracle would have produced."""
import ctypes as C
import os
import sys

import numpy as np

import oracle
from synth_tools_amd import synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _tab(orc):
    return np.array([orc.orc_note_to_inc(n) for n in range(128)], np.uint32)


def test_saw_bus_closed_form_equals_oracle(orc):
    tab = _tab(orc)
    for n, frac in ((5000, 0.8), (64, 1.0), (300, 0.0)):
        inc, state = synthetic.saw_bank(n, 0x5EED0B00 + n, tab, active_fraction=frac)
        inc[:16] = np.uint32(0x7FFFFFF0)                      # loud voices: the int32 sum wraps
        st = state.copy()
        t = 0
        for nf in (1, 7, 64, 130):
            obus, _ = oracle.synth_run(orc, inc, st, nf, want_vec=False)
            got = bench.saw_bus_closed_form(inc, state, t, list(range(nf)))
            assert np.array_equal(got, obus.astype(np.int64)), (n, nf)
            t += nf
    # elapsed counts beyond 2^32 wrap like the phases
    got = bench.saw_bus_closed_form(inc, state, (1 << 32) + 5, [0])
    want = bench.saw_bus_closed_form(inc, state, 5, [0])
    assert np.array_equal(got, want)


def test_pdm_rows_closed_form_equals_oracle(orc):
    n = 4096
    sp, ac = synthetic.pdm_bank(n, 0x5EED0B03)
    ac = (synthetic.splitmix64(5, n) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    nt = 300
    for dither in (None, synthetic.dither_stream(nt, 7, 0x0FFFFFFF)):
        a = ac.copy()
        bits = oracle.pdm_run(orc, sp, a, nt, dither)
        rows = [0, 1, 31, 32, 150, 299]
        want = bench.pdm_rows_closed_form(sp, ac, dither, rows)
        for t, w in zip(rows, want):
            assert np.array_equal(bits[t], w), (t, dither is not None)


def test_poly_block_numpy_equals_oracle(orc):
    tab = _tab(orc)
    n = 3000
    a = synthetic.poly_bank(n, 0x5EED0B04, tab, active_fraction=0.9)
    for blk in range(6):
        if blk == 3:
            a["gate"] = (1 - a["gate"]).astype(np.uint32)      # releases and new attacks
        want_np = bench.poly_block_numpy(a, 64)
        b = oracle.PolyBank(n=n, **{k: v.ctypes.data for k, v in a.items()})
        bus = np.zeros(2 * 64, np.int32)
        orc.orc_poly_run(C.byref(b), bus, 64)                    # advances a[] in place
        assert np.array_equal(want_np, bus.reshape(64, 2).astype(np.int64)), blk


def test_pwm2_block_numpy_equals_oracle(orc):
    """bench.py's PWM leg check: the numpy statement of the firmware's order-2 channel against orc_pwm_bank_run
    (whose shaper is pinned by the real pdm.h), across control-rate boundaries."""
    n = 700
    r = synthetic.splitmix64(0x5EED0B08, 7 * n).reshape(7, n)
    st = {k: (r[i] & np.uint64(0xFFFFFFFF)).astype(np.uint32)
          for i, k in enumerate(("setpoint", "pos0", "vel0", "pos1", "vel1", "s1", "s2"))}
    st["vel0"] = (st["vel0"] >> np.uint32(12)).astype(np.uint32)
    st["vel1"] = (0 - (st["vel1"] >> np.uint32(13))).astype(np.uint32)          # negative velocities too
    for div_log, div_count, nt in ((12, 4096 - 100, 256), (4, 3, 100), (12, 0, 5)):
        o = {k: v.copy() for k, v in st.items()}
        b = oracle.PwmBank(n=n, setpoint=o["setpoint"].ctypes.data, pos0=o["pos0"].ctypes.data, vel0=o["vel0"].ctypes.data,
                           pos1=o["pos1"].ctypes.data, vel1=o["vel1"].ctypes.data,
                           s=(C.c_void_p * 4)(o["s1"].ctypes.data, o["s2"].ctypes.data, None, None), order=2,
                           div_count=div_count, div_log=div_log, out_shift=24)
        dith = synthetic.dither_stream(nt, 11, 0x3FF)
        want = np.zeros((nt, n), np.uint8)
        orc.orc_pwm_bank_run(C.byref(b), dith.ctypes.data, nt, want.ctypes.data)
        got, after = bench.pwm2_block_numpy(st, dith, nt, div_count, div_log=div_log)
        assert np.array_equal(got, want), (div_log, div_count)
        for k in o:
            assert np.array_equal(after[k], o[k]), (k, div_log, div_count)
        st = after


def test_roof_and_wrap_helpers():
    assert bench.wrap_i32([2**31, -2**31 - 1, 5]).tolist() == [-2**31, 2**31 - 1, 5]
    r = bench.roof(8e9, 1.0)                                    # 8 GB in 1 ms = 8 TB/s
    assert abs(r["hbm_frac"] - 1.0) < 1e-3 and r["bound"] == "hbm"
    r = bench.roof(1.0, 1.0, units=64 * bench.SIMD_CYCLES_PER_S * 1e-3, issue_cycles=1.0)
    assert abs(r["valu_issue_frac"] - 1.0) < 1e-3 and r["bound"] == "vector issue"
