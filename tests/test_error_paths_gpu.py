"""GPU: argument errors and empty inputs through the raw C-ABI.  The reference's handlers return
0 / -1 / -2 / -3 (mod_synth.c:91, 106-107, 113); the bank API keeps that style: SMX_E_ARG = -1 for a
bad argument, SMX_E_RANGE = -2 for an index out of range, SMX_E_STATE = -3 for a call out of order,
NULL + smx_last_error() from constructors.  A failed call must leave the object usable."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

E_ARG, E_RANGE, E_STATE = -1, -2, -3


def _err(L):
    return L.smx_last_error().decode()


def test_constructors_reject_bad_sizes_and_devices(smx):
    L = smx.lib()
    ndev = L.smx_device_count()
    assert ndev >= 1
    for create, args in ((L.smx_bank_create, (0, 0)), (L.smx_bank_create, (100, ndev)), (L.smx_bank_create, (100, -1)),
                         (L.smx_bank_create, (0xFFFFF001, 0)),
                         (L.smx_pdm_create, (0, 0)), (L.smx_pdm_create, (8, ndev)),
                         (L.smx_poly_create, (0, 0)), (L.smx_osc_create, (0, 0)), (L.smx_clock_create, (0, 0)),
                         (L.smx_pwm_create, (8, 0, 0)), (L.smx_pwm_create, (8, 5, 0)), (L.smx_pwm_create, (0, 2, 0))):
        assert not create(*args), (create.__name__, args)
        assert _err(L), create.__name__
    # destroying NULL is a no-op everywhere
    for destroy in (L.smx_bank_destroy, L.smx_pdm_destroy, L.smx_poly_destroy, L.smx_pwm_destroy,
                    L.smx_osc_destroy, L.smx_clock_destroy, L.smx_cproc_destroy, L.smx_fw_destroy):
        destroy(None)


def test_saw_bank_bad_calls_leave_it_usable(smx, orc, inc_table):
    import oracle
    from synth_tools_amd import synthetic
    L = smx.lib()
    n = 777
    inc, st = synthetic.saw_bank(n, 5, inc_table)
    bank = smx.SawBank(n)
    bank.load(inc, st)
    h = bank._h
    vec = np.zeros(64, np.float32)
    bus = np.zeros(64, np.int32)
    R = C.CDLL(smx.LIB_PATH)                                     # raw handle: NULL where the table wants arrays
    assert L.smx_bank_run(h, vec.ctypes.data, bus.ctypes.data, 0) == E_ARG
    assert L.smx_bank_run(h, vec.ctypes.data, bus.ctypes.data, -5) == E_ARG
    assert L.smx_bank_run(None, vec.ctypes.data, bus.ctypes.data, 64) == E_ARG
    assert L.smx_bank_run_async(h, 0) == E_ARG
    assert L.smx_bank_note_on(h, -1) == E_ARG and L.smx_bank_note_off(h, -1) == E_ARG
    assert L.smx_bank_note_on(None, 60) == E_ARG
    assert L.smx_bank_set_block_mode(h, 7) == E_ARG
    assert L.smx_bank_allreduce_async(h, 64) == E_STATE          # no communicator yet
    assert L.smx_bank_comm_init(h, 2, 2, np.zeros(smx.UNIQUE_ID_BYTES, np.uint8)) == E_ARG   # rank >= nranks
    assert R.smx_bank_midi_events(C.c_void_p(h), None, C.c_size_t(3)) == E_ARG
    assert R.smx_bank_midi_events(C.c_void_p(h), None, C.c_size_t(0)) == 0      # an empty block of events
    assert L.smx_bank_midi_event(h, np.zeros(3, np.uint8), 0) == 0    # sizes other than 3 are ignored (linux/synth.c:236)
    assert L.smx_bank_load(h, None, None) == 0                   # nothing to replace
    assert L.smx_bank_voices(None) == 0 and L.smx_bank_voices(h) == n
    assert _err(L)
    # ... and the bank still renders the right samples
    got, _ = bank.run(64)
    want, _ = oracle.synth_run(orc, inc, st, 64)
    assert np.array_equal(got, want)
    bank.close()


def test_pdm_pwm_poly_bad_calls(smx, orc):
    import oracle
    from synth_tools_amd import synthetic
    L = smx.lib()
    n = 100
    sp, ac = synthetic.pdm_bank(n, 3)
    p = smx.PdmBank(n)
    p.load(sp, ac)
    assert L.smx_pdm_set_setpoint(p._h, n, 5) == E_RANGE         # mod_synth.c:106: bad channel -> -2
    assert L.smx_pdm_set_setpoint(p._h, 0xFFFFFFFF, 5) == E_RANGE
    assert L.smx_pdm_set_setpoint(None, 0, 5) == E_ARG
    assert L.smx_pdm_tick_n_streams(p._h, 33, None, None) == E_ARG   # whole 32-tick words only
    assert L.smx_pdm_tick_n(p._h, 0, None, None) == 0            # zero ticks: nothing happens
    gsp, gac = p.read()
    assert np.array_equal(gsp, sp) and np.array_equal(gac, ac)
    oa = ac.copy()
    assert np.array_equal(p.tick_n(70), oracle.pdm_run(orc, sp, oa, 70, None))
    p.close()

    w = smx.PwmBank(n, order=2)
    assert L.smx_pwm_set_setpoint(w._h, n, 1) == E_RANGE
    assert L.smx_pwm_set_div_count(w._h, 1 << 12) == E_ARG       # div_count < 1 << div_log
    w.close()

    pb = smx.PolyBank(n)
    assert L.smx_poly_run_async(pb._h, 0) == E_ARG
    assert L.smx_poly_run_async(pb._h, 65) == E_ARG              # at most 64 frames per launch
    assert L.smx_poly_run(pb._h, None, None, 0) == E_ARG
    bus, _ = pb.run(130)                                          # all voices off: silence, any length
    assert not bus.any()
    pb.close()

    o = smx.OscBank(n)
    assert o.set_log_max(0) == E_ARG and o.set_log_max(32) == E_ARG and o.set_log_max(26) == 0   # pmeas shift range
    assert L.smx_osc_tick_n(o._h, 0, None, None) == 0
    assert L.smx_osc_events(o._h, 0, None, None) == 0
    assert L.smx_osc_events(o._h, 3, None, None) == E_ARG
    o.close()


def test_published_bus_poll_is_bounded(smx):
    """smx_bank_fetch polls the sequence word that the stream's last kernel writes to pinned host memory; the poll must
    never spin forever (a lost device on a real-time thread).  With SMX_PUBLISH_TIMEOUT_MS=0 and a launch of a few
    hundred microseconds the call returns SMX_E_NOGPU with a message instead of the bus, the process goes on, and
    once the stream has drained the bank is intact (the block DID run: the phases have advanced by its frames)."""
    import os
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        import numpy as np
        import synth_tools_amd as sta
        L = sta.lib()
        n = 1 << 26
        inc = np.full(n, 12345, np.uint32); st = np.zeros(n, np.uint32)
        b = sta.SawBank(n); b.load(inc, st)
        vec = np.zeros(64, np.float32)
        rv = L.smx_bank_run(b._h, vec.ctypes.data, None, 64)
        msg = L.smx_last_error().decode()
        assert rv == -10, rv                                  # SMX_E_NOGPU
        assert "did not publish" in msg, msg
        b.sync()
        ginc, gst = b.read()
        assert (gst == np.uint32(64 * 12345)).all()
        print("bounded")
    """) % root
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SMX_PUBLISH_TIMEOUT_MS="0"),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "bounded" in p.stdout, p.stderr[-2000:]
