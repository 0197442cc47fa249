"""GPU: the entry points that hand out DEVICE pointers (for a caller that keeps its inputs / outputs in HBM:
smx_bank_bus_dev, smx_pdm_dither_dev, smx_pdm_bits_dev, smx_pwm_dither_dev), the state loaders that mirror a read
(smx_cproc_load_state, smx_osc_load_pmeas) and smx_version -- every exported function is called by some test.
The pointers are used the way such a caller would: plain hipMemcpy on them (libamdhip64 through ctypes), after the
bank's own sync call."""
import ctypes as C
import re
import os

import numpy as np
import pytest

import oracle
from synth_tools_amd import synthetic

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H2D, D2H = 1, 2


@pytest.fixture(scope="module")
def hip(smx):
    h = C.CDLL("libamdhip64.so")          # already mapped by the product library: the same runtime
    h.hipMemcpy.restype = C.c_int
    h.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def copy(dst, src, nbytes, kind):
        assert h.hipMemcpy(dst, src, nbytes, kind) == 0
    return copy


def _dev(fn, *args):
    fn.restype = C.c_void_p
    p = fn(*args)
    assert p, "null device pointer"
    return p


def test_version_matches_the_header(smx):
    hdr = open(os.path.join(ROOT, "include", "synth_mi355x.h")).read()
    m = re.search(r"#define\s+SMX_VERSION\s+(\w+)", hdr)
    assert m, "the header defines SMX_VERSION"
    assert smx.lib().smx_version() == int(m.group(1), 0)


@pytest.mark.parametrize("n", [1000, 65536, (1 << 20) + 4096])
def test_bank_bus_dev_is_the_last_blocks_bus(smx, orc, inc_table, hip, n):
    """The pointer taken AFTER run_async holds that block's int32 bus once the bank is synced -- also where the
    launch left its slot fold to a successor (>= 2^20 voices, direct form)."""
    inc, state = synthetic.saw_bank(n, 0x5EED0B00 + n, inc_table, active_fraction=0.8)
    bank = smx.SawBank(n)
    bank.load(inc, state)
    st = state.copy()
    for nf in (64, 1, 16, 200, 33):
        bank.run_async(nf)
        p = _dev(smx.lib().smx_bank_bus_dev, bank._h)
        bank.sync()
        got = np.empty(nf, np.int32)
        hip(got.ctypes.data, p, nf * 4, D2H)
        want, _ = oracle.synth_run(orc, inc, st, nf)
        assert np.array_equal(got, want), (n, nf)
        assert np.array_equal(bank.fetch(nf)[0], want)
    bank.close()


@pytest.mark.parametrize("n", [70, 5000])
def test_pdm_dither_and_bits_stay_in_hbm(smx, orc, hip, n):
    """smx_pdm_dither_dev filled by the caller, smx_pdm_tick_n_async(with_dither), smx_pdm_bits_dev read by the
    caller == smx_pdm_tick_n with host arrays (and the oracle)."""
    sp, accu = synthetic.pdm_bank(n, 0x5EED0B10 + n)
    bank = smx.PdmBank(n)
    bank.load(sp, accu)
    oa = accu.copy()
    L = smx.lib()
    words = (n + 31) // 32
    row_words = ((n + 1023) // 1024) * 1024 // 32            # the tick-major matrix is padded to 1024 channels per row
    for k, nt in enumerate([64, 3, 200]):
        d = synthetic.dither_stream(nt, 77 + k, 0x0FFFFFFF)
        pd = _dev(L.smx_pdm_dither_dev, bank._h, nt)
        hip(pd, d.ctypes.data, nt * 4, H2D)
        bank.tick_n_async(nt, with_dither=True)
        bank.sync()
        pb = _dev(L.smx_pdm_bits_dev, bank._h)
        raw = np.empty((nt, row_words), np.uint32)
        hip(raw.ctypes.data, pb, raw.nbytes, D2H)
        want = oracle.pdm_run(orc, sp, oa, nt, d)
        assert np.array_equal(raw[:, :words], want), (n, nt)
    assert np.array_equal(bank.read()[1], oa)
    bank.close()


def test_pwm_dither_in_hbm(smx, orc, hip):
    """smx_pwm_dither_dev + smx_pwm_tick_n_async(with_dither) leave the bank where smx_pwm_tick_n(dither) does."""
    n = 3000
    r = synthetic.splitmix64(0x5EED0B20, 5 * n).reshape(5, n)
    u = lambda k: (r[k] >> np.uint64(32)).astype(np.uint32)
    arrs = dict(setpoint=u(0), pos0=u(1), vel0=(u(2) >> np.uint32(12)) - np.uint32(1 << 19), pos1=u(3),
                vel1=(u(4) >> np.uint32(12)) - np.uint32(1 << 19))
    a = smx.PwmBank(n, order=2, control_div_log=5)
    b = smx.PwmBank(n, order=2, control_div_log=5)
    a.load(**arrs)
    b.load(**arrs)
    L = smx.lib()
    for k, nt in enumerate([40, 1, 100]):
        d = synthetic.dither_stream(nt, 90 + k, 0x3FF)
        pd = _dev(L.smx_pwm_dither_dev, a._h, nt)
        hip(pd, d.ctypes.data, nt * 4, H2D)
        a.tick_n_async(nt, with_dither=True)
        a.sync()
        b.tick_n(nt, d, want_duty=False)
        assert a.div_count == b.div_count
    sa, sb = a.read(), b.read()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    a.close()
    b.close()


def test_cproc_load_state_resumes_a_chain(smx, orc):
    """read_state -> a fresh bank -> load_state: the second bank goes on exactly where the first one was."""
    from synth_tools_amd import PROC_ACC, PROC_EDGE, cproc_input
    nodes = [(PROC_EDGE, cproc_input(0), 1), (PROC_ACC, 0, 1), (PROC_ACC, 1, 3)]
    n = 777
    rng = np.random.default_rng(0xB30)
    inp1 = rng.integers(0, 3, (40, 1, n)).astype(np.uint32)
    inp2 = rng.integers(0, 3, (40, 1, n)).astype(np.uint32)
    a = smx.CprocBank(n, nodes, 1)
    a.tick_n(inp1)
    state = a.read_state()
    want = a.tick_n(inp2)
    b = smx.CprocBank(n, nodes, 1)
    st = np.ascontiguousarray(state, np.uint32)
    assert smx.lib().smx_cproc_load_state(b._h, st.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(b.read_state(), state)
    assert np.array_equal(b.tick_n(inp2), want)
    assert np.array_equal(b.read_state(), a.read_state())
    a.close()
    b.close()


def test_osc_load_pmeas_resumes_a_measurement(smx, orc):
    """read_pmeas -> a fresh bank -> load_pmeas: the period measurement (pmeas.h:64-108) goes on from that state."""
    n, log_max = 300, 16
    rng = np.random.default_rng(0xB40)
    period = rng.integers(200, 70000, n)

    def events(now, ne):
        cc = np.zeros((ne, n), np.uint32)
        for e in range(ne):
            now = now + (period + rng.integers(0, 50, n)).astype(np.uint64)
            cc[e] = (now & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        return cc, now
    a = smx.OscBank(n)
    assert a.set_log_max(log_max) == 0
    cc1, now = events(rng.integers(0, 2**32, n, dtype=np.uint64), 50)
    cc2, _ = events(now, 50)
    a.events(cc1)
    mid = a.read_pmeas()
    a.events(cc2)
    want = a.read_pmeas()
    b = smx.OscBank(n)
    assert b.set_log_max(log_max) == 0
    b.load_pmeas(**mid)
    got_mid = b.read_pmeas()
    for k in mid:
        assert np.array_equal(got_mid[k], mid[k]), k
    b.events(cc2)
    got = b.read_pmeas()
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    assert int(want["write"].max()) >= 2
    a.close()
    b.close()


def test_create_destroy_leaves_no_device_memory_behind(smx, inc_table):
    """Every bank kind created, used (so that its lazily allocated buffers exist: bus ring, scratch, staging, pulse and
    duty matrices, pinned words) and destroyed, fifty times over: hipMemGetInfo's free figure is back where it
    started (within the allocator's granularity)."""
    h = C.CDLL("libamdhip64.so")
    h.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert h.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    from synth_tools_amd import PROC_ACC, PROC_EDGE, PROC_GPIN, cproc_input
    n = (1 << 20) + 4096

    def cycle():
        inc, st = synthetic.saw_bank(n, 0x5EED0D00, inc_table)
        b = smx.SawBank(n)
        b.load(inc, st)
        b.run(64); b.run_async(300); b.run_async(5000); b.fetch(64)
        b.midi_events(np.array([[0x90, 60, 64], [0x80, 60, 0]], np.uint8))
        b.set_block_mode(True); b.run(64); b.run(64)
        b.close()
        p = smx.PdmBank(n)
        p.init(); p.tick_n(70); p.tick_n(2); p.tick_n_streams(64)
        p.close()
        w = smx.PwmBank(1 << 16)
        w.init(); w.tick_n(40, synthetic.dither_stream(40, 1, 0x3FF))
        w.close()
        o = smx.OscBank(5000)
        o.tick_n(33); o.events(np.zeros((2, 5000), np.uint32))
        o.close()
        y = smx.PolyBank(1 << 16)
        y.run(64); y.run_async(7)
        y.close()
        c = smx.CprocBank(3000, [(PROC_EDGE, cproc_input(0), 1), (PROC_ACC, 0, 1)], 1)
        c.tick_n(np.zeros((8, 1, 3000), np.uint32))
        c.close()
        t = smx.Patch(3000, 1)
        t.apply(PROC_GPIN, [], 0); t.apply(PROC_ACC, [0]); t.tick(3, np.zeros((3, 1, 3000), np.uint32))
        t.close()
        m = smx.ModPdm(70, 3)
        m.tick_n(300)
        m.close()
        f = smx.Firmware()
        f.tick_n(10)
        f.close()
        k = smx.ClockBank(100)
        k.run(64)
        k.close()

    def host_side():
        rss_pages = int(open("/proc/self/statm").read().split()[1])
        return rss_pages * os.sysconf("SC_PAGE_SIZE"), len(os.listdir("/proc/self/fd"))

    cycle()                                                  # the runtime's own pools settle
    cycle()
    before = free_bytes()
    rss0, fds0 = host_side()
    for _ in range(50):
        cycle()
    after = free_bytes()
    rss1, fds1 = host_side()
    print("host side: rss %+d MiB, open descriptors %+d after 50 cycles" % ((rss1 - rss0) >> 20, fds1 - fds0))
    assert fds1 - fds0 <= 2, "descriptors leak: %d -> %d" % (fds0, fds1)            # streams / events / pinned mappings
    assert rss1 - rss0 < (256 << 20), "resident set grew by %d MiB" % ((rss1 - rss0) >> 20)   # pinned buffers are resident
    print("free device memory: %d KiB less after 50 cycles" % ((before - after) >> 10))
    assert before - after < (8 << 20), "device memory shrank by %d KiB over 50 create/destroy cycles" % ((before - after) >> 10)
