"""Seeded synthetic banks: splitmix64, vectorised with numpy.

Host-side input generation only; used by tests and bench.py so that the GPU
path, the oracle and the CPU baseline all see the same bytes.
"""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, n):
    """n outputs of splitmix64 started at `seed` (uint64 array)."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * np.arange(1, n + 1, dtype=np.uint64)) & _M
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
        return z ^ (z >> np.uint64(31))


def note_inc_table(note_to_inc):
    return np.array([note_to_inc(n) for n in range(128)], np.uint32)


def saw_bank(n, seed, inc_table, active_fraction=1.0):
    """c2/c5 distribution: inc = note_to_inc(21 + r % 88) (piano range), state = r32."""
    r = splitmix64(seed, n)
    inc = inc_table[21 + (r % np.uint64(88)).astype(np.int64)].astype(np.uint32)
    state = (r >> np.uint64(32)).astype(np.uint32)
    if active_fraction < 1.0:
        off = (splitmix64(seed ^ 0xA5A5, n) % np.uint64(1 << 20)).astype(np.float64) / (1 << 20)
        inc = np.where(off < active_fraction, inc, 0).astype(np.uint32)
    return np.ascontiguousarray(inc), np.ascontiguousarray(state)


def pdm_bank(n, seed):
    """c3 distribution: setpoint uniform in the 25-75 % safe range (mod_pdm.c:99-100)."""
    r = splitmix64(seed, n)
    sp = (np.uint64(0x40000000) + (r % np.uint64(0x80000001))).astype(np.uint32)
    accu = np.zeros(n, np.uint32)
    return sp, accu


def dither_stream(nticks, seed, mask):
    return (splitmix64(seed, nticks) & np.uint64(mask)).astype(np.uint32)


def poly_bank(n, seed, inc_table, active_fraction=1.0):
    """c4 distribution: saw bank + LPF coefficient uniform [0.01, 0.5], ADSR rates
    log-uniform, sustain uniform, pan uniform, envelopes at rest, gates random."""
    inc, phase = saw_bank(n, seed, inc_table, active_fraction)
    r = splitmix64(seed ^ 0xC4C4C4C4, 6 * n).reshape(6, n)
    u = lambda k: (r[k] >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    a = (0.01 + 0.49 * u(0)).astype(np.float32)
    # attack 1 ms .. 1 s at 48 kHz: rate = 2^32 / samples
    rate = lambda x: np.minimum(2.0 ** 32 / (48.0 * 10.0 ** (3.0 * x)), 2.0 ** 32 - 1).astype(np.uint64).astype(np.uint32)
    ar, dr, rr = rate(u(1)), rate(u(2)), rate(u(3))
    sl = (u(4) * (2.0 ** 32 - 1)).astype(np.uint64).astype(np.uint32)
    pl = (u(5) * 256.0 + 0.5).astype(np.uint32)
    pan = (pl | ((np.uint32(256) - pl) << np.uint32(16))).astype(np.uint32)
    gate = ((r[0] >> np.uint64(7)) & np.uint64(1)).astype(np.uint32)
    z = np.zeros(n, np.uint32)
    return dict(inc=inc, phase=phase, y=np.zeros(n, np.float32), a=a, level=z.copy(), stage=z.copy(),
                gate=gate, ar=ar, dr=dr, sl=sl, rr=rr, pan=pan)
