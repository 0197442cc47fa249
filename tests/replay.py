"""Replays the scripted sequences of tests/golden/synth_c_reference.npz (rows [op, a, b]; the
generator is tests/golden/make_golden.py) on the oracle, on the compiled reference, or on the
product's drop-in entry points.  Test helper only."""
import ctypes as C

import numpy as np

OP_ON, OP_OFF, OP_RUN, OP_SQUARE, OP_POKE = 0, 1, 2, 3, 4
SCRIPTS = ("b1_ticks", "b64_quirks", "b4096_random", "wrapping_mix", "square")


def on_oracle(orc, script):
    """-> (float bits, note2voice, inc, state) from the oracle's SoA restatement (64 voices)."""
    import oracle
    n2v = np.zeros(128, np.int32)
    inc = np.zeros(64, np.uint32)
    st = np.zeros(64, np.uint32)
    out = []
    for op, a, b in script:
        a = int(a)
        if op == OP_ON:
            orc.orc_note_on(n2v, inc, 64, a)
        elif op == OP_OFF:
            orc.orc_note_off(n2v, inc, 64, a)
        elif op == OP_RUN:
            out.append(oracle.synth_run(orc, inc, st, a)[1])
        elif op == OP_SQUARE:
            out.append(np.array([orc.orc_sum_tick_square(inc, st, 64) for _ in range(a)], np.float32))
        elif op == OP_POKE:
            st[a] = int(b)
    return np.concatenate(out).view(np.uint32), n2v, inc, st


def on_struct_synth(lib, synth_cls, script, square=None):
    """-> the same tuple from anything that exports the reference's own entry points on the
    reference's 1024-byte `struct synth` (linux/synth.c:37-45): the compiled reference
    (oracle/_ref/libref_synth.so) or the product's drop-in (libsynth_mi355x.so).
    `square(x_ptr) -> float` runs one sum_tick_square."""
    x = synth_cls()
    lib.synth_init(C.byref(x))
    out = []
    for op, a, b in script:
        a = int(a)
        if op == OP_ON:
            lib.synth_note_on(C.byref(x), a)
        elif op == OP_OFF:
            lib.synth_note_off(C.byref(x), a)
        elif op == OP_RUN:
            v = np.zeros(a, np.float32)
            lib.synth_run(C.byref(x), v, a)          # both bindings take an ndarray here
            out.append(v)
        elif op == OP_SQUARE:
            out.append(np.array([square(C.byref(x)) for _ in range(a)], np.float32))
        elif op == OP_POKE:
            x.voice[a].note_state = int(b)
    v = np.frombuffer(bytes(x.voice), np.uint32).reshape(64, 2)
    return (np.concatenate(out).view(np.uint32), np.array(x.note2voice[:], np.int32),
            v[:, 0].copy(), v[:, 1].copy())
