"""GPU parity: the HIP carry-out PDM bank (through the C-ABI) against the CPU
oracle.  Bar: bit-exact pulse words and accumulators."""
import json
import os

import numpy as np
import pytest

import oracle
from synth_tools_amd import synthetic

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("n", [1, 2, 3, 31, 32, 33, 64, 65, 1023, 1024, 1025, 5000])
@pytest.mark.parametrize("with_dither", [False, True])
def test_parity_ragged(smx, orc, n, with_dither):
    sp, accu = synthetic.pdm_bank(n, 0x5EED0003 + n)
    accu = (synthetic.splitmix64(n, n) >> np.uint64(32)).astype(np.uint32)
    bank = smx.PdmBank(n)
    bank.load(sp, accu)
    oa = accu.copy()
    for k, nt in enumerate([1, 2, 63, 64, 65, 128, 200]):
        d = synthetic.dither_stream(nt, 1000 + k, 0x0FFFFFFF) if with_dither else None   # mod_pdm.c:261
        got = bank.tick_n(nt, d)
        want = oracle.pdm_run(orc, sp, oa, nt, d)
        assert np.array_equal(got, want), "n=%d nt=%d" % (n, nt)
    gsp, gac = bank.read()
    assert np.array_equal(gsp, sp) and np.array_equal(gac, oa)
    bank.close()


def test_reference_two_channel_config(smx, orc):
    """pdm_init defaults (mod_pdm.c:320-326), 2 channels (mod_pdm.c:124-125), BSRR words."""
    kat = json.load(open(os.path.join(GOLD, "survey_known_answers.json")))["mod_pdm_two_channel_derived"]
    bank = smx.PdmBank(2)
    bank.init()
    sp, ac = bank.read()
    assert sp.tolist() == kat["setpoint"] and ac.tolist() == [0, 0]
    bits = bank.tick_n(kat["ticks"])
    L = smx.lib()
    assert [L.smx_pdm_bsrr_word(int(w), 2) for w in bits[:, 0]] == kat["bsrr"]
    assert bank.read()[1].tolist() == kat["accu_end"]
    # SETPOINT command (mod_synth.c:104-111): range check -> -2
    assert bank.set_setpoint(2, 5) == -2
    assert bank.set_setpoint(1, 0x80000000) == 0
    assert bank.read()[0].tolist() == [2000000000, 0x80000000]
    bank.close()


def test_comment_kat_on_gpu(smx):
    """stm32f103/mod_pdm.c:43-47, X=3 on a 3-bit accumulator, scaled by 2^29."""
    k = json.load(open(os.path.join(GOLD, "survey_known_answers.json")))["mod_pdm_comment_kat_3bit"]
    bank = smx.PdmBank(1)
    bank.load(np.array([k["X"] << 29], np.uint32), np.array([5 << 29], np.uint32))
    bits = bank.tick_n(len(k["C"]))
    assert bits[:, 0].tolist() == k["C"]
    bank.close()


def test_comment_kat_x5_accumulator_row_on_gpu(smx):
    """mod_pdm.c:49-53: A: 0 5 2 7 4 1 6 3 0 for X = 5 on a 3-bit accumulator (scaled by 2^29), read back
    with smx_pdm_read after every tick; and "the same waveform, but in reverse" against the X = 3 row."""
    kat = json.load(open(os.path.join(GOLD, "survey_known_answers.json")))
    k = kat["mod_pdm_comment_kat_3bit_x5"]
    bank = smx.PdmBank(1)
    bank.load(np.array([k["X"] << 29], np.uint32), np.array([k["A"][0] << 29], np.uint32))
    carries = []
    for want in k["A"][1:]:
        carries.append(int(bank.tick_n(1)[0, 0]))
        assert int(bank.read()[1][0]) == want << 29
    bank.close()
    assert [1 - c for c in carries] == kat["mod_pdm_comment_kat_3bit"]["C"][1:][::-1]


def test_comment_period_table_on_gpu(smx):
    """mod_pdm.c:30-38: X = 2^k -> one pulse every 2^(32-k) ticks; all testable rows as channels of ONE bank
    (plus X = 1 with the accumulator three ticks before its wrap), 12 288 ticks in ragged calls."""
    lgs = json.load(open(os.path.join(GOLD, "survey_known_answers.json")))["mod_pdm_comment_period_table"]["testable_log2_X_32bit"]
    sp = np.array([1 << lg for lg in lgs] + [1], np.uint32)
    ac = np.array([0] * len(lgs) + [0xFFFFFFFD], np.uint32)
    bank = smx.PdmBank(len(sp))
    bank.load(sp, ac)
    nt = 3 * 4096
    bits = np.concatenate([bank.tick_n(k) for k in (1, 63, 64, 4000, nt - 4128)])[:, 0]
    for c, lg in enumerate(lgs):
        period = 1 << (32 - lg)
        assert np.flatnonzero((bits >> np.uint32(c)) & 1).tolist() == list(range(period - 1, nt, period)), lg
    assert np.flatnonzero((bits >> np.uint32(len(lgs))) & 1).tolist() == [2]
    assert bank.read()[1].tolist() == [(nt << lg) & 0xFFFFFFFF for lg in lgs] + [nt - 3]
    bank.close()


def test_c3_full_size_1m_channels(smx, orc):
    """BASELINE config 3: 1 Mi channels.  Size-independent properties with dither = 0:
    accu' = T*setpoint mod 2^32 and pulses(channel) = floor(T*setpoint / 2^32) exactly;
    plus a 2048-channel slice against the oracle bit for bit."""
    n, nt = 1 << 20, 4096
    sp, accu = synthetic.pdm_bank(n, 0x5EED0003)
    bank = smx.PdmBank(n)
    bank.load(sp, accu)
    bits = bank.tick_n(nt)
    _, gac = bank.read()
    total = sp.astype(np.uint64) * np.uint64(nt)
    assert np.array_equal(gac, (total & np.uint64(0xFFFFFFFF)).astype(np.uint32))
    # pulse count per channel for a sample of channels, via bit-plane sums
    cols = np.r_[0:64, n // 2:n // 2 + 64, n - 64:n]
    for c in cols:
        ones = int(((bits[:, c >> 5] >> np.uint32(c & 31)) & 1).sum())
        assert ones == int(total[c] >> np.uint64(32))
    lo = 5 * 1024
    sl = slice(lo, lo + 2048)
    oa = np.zeros(2048, np.uint32)
    want = oracle.pdm_run(orc, np.ascontiguousarray(sp[sl]), oa, nt)
    assert np.array_equal(bits[:, lo // 32:(lo + 2048) // 32], want)
    bank.close()


def test_c3_full_size_with_dither(smx, orc):
    """1 Mi channels x 2048 ticks with a seeded dither stream (mod_pdm.c:261 mask): closed-form
    accumulators and exact pulse counts, accu' = sum_t ((sp + d_t) mod 2^32) mod 2^32 and
    pulses = floor(that sum / 2^32), plus a slice against the oracle."""
    n, nt = 1 << 20, 2048
    sp, accu = synthetic.pdm_bank(n, 0x5EED0033)
    d = synthetic.dither_stream(nt, 0xD17D17, 0x0FFFFFFF)
    bank = smx.PdmBank(n)
    bank.load(sp, accu)
    bits = bank.tick_n(nt, d)
    _, gac = bank.read()
    cols = np.r_[0:32, n // 3:n // 3 + 32, n - 32:n]
    x = (sp[cols].astype(np.uint64)[None, :] + d.astype(np.uint64)[:, None]) & np.uint64(0xFFFFFFFF)
    total = x.sum(axis=0)
    assert np.array_equal(gac[cols], (total & np.uint64(0xFFFFFFFF)).astype(np.uint32))
    for k, c in enumerate(cols):
        ones = int(((bits[:, c >> 5] >> np.uint32(c & 31)) & 1).sum())
        assert ones == int(total[k] >> np.uint64(32))
    lo = 9 * 1024
    oa = np.zeros(1024, np.uint32)
    want = oracle.pdm_run(orc, np.ascontiguousarray(sp[lo:lo + 1024]), oa, nt, d)
    assert np.array_equal(bits[:, lo // 32:(lo + 1024) // 32], want)
    assert np.array_equal(gac[lo:lo + 1024], oa)
    bank.close()


def _to_streams(bits, n):
    """tick-major pulse words -> channel streams [ticks/32][n] (numpy reference transform)."""
    nt = bits.shape[0]
    b = ((bits[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(nt, -1)[:, :n].astype(np.uint32)   # [tick][channel]
    b = b.reshape(nt // 32, 32, n)
    return (b << np.arange(32, dtype=np.uint32)[None, :, None]).sum(axis=1, dtype=np.uint64).astype(np.uint32)


@pytest.mark.parametrize("n", [1, 5, 1023, 1025, 3000])
@pytest.mark.parametrize("with_dither", [False, True])
def test_channel_stream_layout(smx, orc, n, with_dither):
    """smx_pdm_tick_n_streams: same pulses as the tick-major matrix, stored per channel."""
    sp, _ = synthetic.pdm_bank(n, 0x5EED0350 + n)
    accu = (synthetic.splitmix64(n + 1, n) >> np.uint64(32)).astype(np.uint32)
    bank = smx.PdmBank(n)
    bank.load(sp, accu)
    oa = accu.copy()
    for k, nt in enumerate([32, 64, 160]):
        d = synthetic.dither_stream(nt, 2000 + k, 0x0FFFFFFF) if with_dither else None
        got = bank.tick_n_streams(nt, d)
        want = _to_streams(oracle.pdm_run(orc, sp, oa, nt, d), n)
        assert np.array_equal(got, want), "n=%d nt=%d" % (n, nt)
        if k == 1:                                   # the two layouts share the accumulators
            got2 = bank.tick_n(40, None)
            assert np.array_equal(got2, oracle.pdm_run(orc, sp, oa, 40, None))
    assert np.array_equal(bank.read()[1], oa)
    with pytest.raises(smx.SmxError):
        bank.tick_n_streams(33)
    bank.close()


def test_quarter_billion_channels_index_safety(smx):
    """2^28 + 1000 channels (the tick-major pulse matrix of 100 ticks is 3.4 GB: offsets beyond
    2^32), dither 0, against closed forms that need no per-tick loop:
      accu after T ticks        = accu0 + T*setpoint                      (mod 2^32)
      pulses of channel c so far = floor((accu0_c + T*setpoint_c) / 2^32)
    so the popcount of tick rows 0..T-1 must equal the sum of those floors (checked at T = 1, 64 and
    100: a full tile, a ragged tile and one tick), and channel c's own pulses of the first 64
    ticks must be its carry sequence (checked on scattered channels incl. the last one)."""
    n = (1 << 28) + 1000
    idx = np.arange(n, dtype=np.uint64)
    sp = (np.uint64(0x40000000) + ((idx * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(33))).astype(np.uint32)
    a0 = ((idx * np.uint64(2654435761)) >> np.uint64(3)).astype(np.uint32)
    del idx
    bank = smx.PdmBank(n)
    bank.load(sp, a0)
    T = 100
    bits = bank.tick_n(T)                                      # (T, words) uint32
    assert bits.shape == (T, (n + 31) // 32)
    row_counts = np.bitwise_count(bits).sum(axis=1, dtype=np.int64)
    for t_end in (1, 64, 100):
        want = np.int64(((a0.astype(np.uint64) + np.uint64(t_end) * sp) >> np.uint64(32)).sum(dtype=np.uint64))
        assert row_counts[:t_end].sum() == want, t_end
    for c in (0, 1, 63, 64, 1023, 1024, (1 << 24) + 5, (1 << 28) - 1, 1 << 28, n - 1):
        acc = np.uint64(a0[c])
        for t in range(64):
            nxt = acc + np.uint64(sp[c])
            pulse = int(nxt >> np.uint64(32))
            acc = nxt & np.uint64(0xFFFFFFFF)
            assert (int(bits[t, c >> 5]) >> (c & 31)) & 1 == pulse, (c, t)
    gsp, gac = bank.read()
    assert np.array_equal(gsp, sp)
    assert np.array_equal(gac, a0 + np.uint32(T) * sp)
    # the tick regime's read-stream kernel at this size (1 tick: per-word stores; 3 ticks: one 32-byte store per tick
    # and wave-trip), from lazily kept accumulators: popcounts of the rows against the same floors
    for nt in (1, 3):
        rows = np.bitwise_count(bank.tick_n(nt)).sum(axis=1, dtype=np.int64)
        acc = (a0.astype(np.uint64) + np.uint64(T) * sp) & np.uint64(0xFFFFFFFF)
        for t in range(nt):
            nxt = acc + sp
            assert rows[t] == np.int64((nxt >> np.uint64(32)).sum(dtype=np.uint64)), (nt, t)
            acc = nxt & np.uint64(0xFFFFFFFF)
        T += nt
    assert np.array_equal(bank.read()[1], a0 + np.uint32(T) * sp)
    bank.close()
