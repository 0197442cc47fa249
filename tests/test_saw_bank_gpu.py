"""GPU parity: the HIP saw bank (through the C-ABI) against the CPU oracle.
Bar: bit-exact (integer bus, final phases, and the float vec, whose only
rounding is int32->float32, linux/synth.c:180)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle
from synth_tools_amd import synthetic

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _check(smx, orc, inc, state, frames_list):
    bank = smx.SawBank(len(inc))
    bank.load(inc, state)
    st = state.copy()
    for nf in frames_list:
        bus, vec = bank.run(nf)
        obus, ovec = oracle.synth_run(orc, inc, st, nf)
        assert np.array_equal(bus, obus), "bus differs at n=%d frames=%d" % (len(inc), nf)
        assert np.array_equal(vec.view(np.uint32), ovec.view(np.uint32))
    ginc, gst = bank.read()
    assert np.array_equal(ginc, inc) and np.array_equal(gst, st)
    bank.close()


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1000, 1024, 1025, 4097])
def test_small_banks_ragged(smx, orc, inc_table, n):
    inc, state = synthetic.saw_bank(n, 0x5EED0000 + n, inc_table, active_fraction=0.7)
    _check(smx, orc, inc, state, [1, 2, 3, 5, 17, 31, 32, 33, 63, 64, 65, 127, 128, 200])


def test_c2_65536_voices(smx, orc, inc_table):
    """BASELINE config 2: 65 536 saw voices, bit-exact vs the CPU loop."""
    inc, state = synthetic.saw_bank(65536, 0x5EED0002, inc_table)
    _check(smx, orc, inc, state, [64, 64, 1, 1024, 100])


def test_vector_path_1m_voices(smx, orc, inc_table):
    """>= 2^20 voices take the 4-voices-per-lane kernel; ragged count, off voices."""
    n = (1 << 20) + 5
    inc, state = synthetic.saw_bank(n, 0x5EED0005, inc_table, active_fraction=0.9)
    _check(smx, orc, inc, state, [64, 7, 1, 130])


def test_all_off_and_all_full_scale(smx, orc):
    n = 2048
    _check(smx, orc, np.zeros(n, np.uint32), np.arange(n, dtype=np.uint32) * 77, [64, 3])
    # every voice near full scale: the 32-bit mix wraps many times (linux/synth.c:171 `int sum`)
    _check(smx, orc, np.full(n, 1, np.uint32), np.full(n, 0x7FFFFFF0, np.uint32), [64])
    _check(smx, orc, np.full(n, 0xFFFFFFFF, np.uint32), np.full(n, 0x80000000, np.uint32), [64, 64])


def test_note_on_off_over_n_voices(smx, orc):
    """Allocator semantics of linux/synth.c:145-165 over a 1000-voice bank, including
    stealing voice 0 when full and stray note-offs."""
    n = 100
    bank = smx.SawBank(n)
    n2v = np.zeros(128, np.int32)
    inc = np.zeros(n, np.uint32)
    st = np.zeros(n, np.uint32)
    rng = np.random.default_rng(11)
    was_full = False
    for step in range(400):
        note = int(rng.integers(0, 128))
        if rng.random() < 0.75:
            was_full = was_full or np.count_nonzero(inc) == n     # this note-on steals voice 0
            bank.note_on(note)
            orc.orc_note_on(n2v, inc, n, note)
        else:
            bank.note_off(note)
            orc.orc_note_off(n2v, inc, n, note)
        if step % 25 == 0:
            bus, _ = bank.run(64)
            obus, _ = oracle.synth_run(orc, inc, st, 64)
            assert np.array_equal(bus, obus)
    ginc, gst = bank.read()
    assert np.array_equal(ginc, inc) and np.array_equal(gst, st)
    assert was_full                              # the bank did fill up: stealing was exercised
    # (not "is full at the end": under SMX_SOAK_SEED a stray note-off may have silenced voice 0 last)
    bank.close()


def test_dropin_synth_run(smx, orc):
    """The reference's own entry points on a caller-owned struct synth
    (linux/synth.c:42-45), incl. the SURVEY A.2 chord known-answer."""
    L = smx.lib()
    kat = json.load(open(os.path.join(GOLD, "survey_known_answers.json")))
    x = smx.Synth()
    L.synth_init(C.byref(x))
    for note in (69, 72, 76):
        L.synth_note_on(C.byref(x), note)
    vec = np.zeros(8, np.float32)
    L.synth_run(C.byref(x), vec, 8)
    assert ["%08x" % v for v in vec.view(np.uint32)] == kat["chord_69_72_76_first8_float_bits"]
    assert x.voice[0].note_inc == kat["chord_voice0_after8"]["inc"]
    assert x.voice[0].note_state == kat["chord_voice0_after8"]["state"]

    g = np.load(os.path.join(GOLD, "synth_run_derived.npz"))
    L.synth_init(C.byref(x))
    vecs = []
    for op, a in g["script"]:
        if op == 0:
            L.synth_midi_event(C.byref(x), np.array([0x90, a, 100], np.uint8), 3)
        elif op == 1:
            L.synth_midi_event(C.byref(x), np.array([0x80, a, 0], np.uint8), 3)
        else:
            v = np.zeros(int(a), np.float32)
            L.synth_run(C.byref(x), v, int(a))
            vecs.append(v)
    assert np.array_equal(np.concatenate(vecs).view(np.uint32), g["vec"].view(np.uint32))
    assert [x.voice[v].note_state for v in range(64)] == g["state"].tolist()
    assert [x.voice[v].note_inc for v in range(64)] == g["inc"].tolist()


def test_square_variant(smx, orc, inc_table):
    """sum_tick_square, linux/synth.c:182-195."""
    for n, frac in ((64, 0.3), (3000, 0.001), (3000, 0.0)):
        inc, state = synthetic.saw_bank(n, 0x5EED0100 + n, inc_table, active_fraction=frac)
        bank = smx.SawBank(n)
        bank.load(inc, state)
        st = state.copy()
        for nf in (64, 1, 100):
            got = bank.run_square(nf)
            want = np.array([orc.orc_sum_tick_square(inc, st, n) for _ in range(nf)], np.float32)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert np.array_equal(bank.read()[1], st)
        bank.close()


def test_full_size_properties_8m_voices(smx, orc, inc_table):
    """BASELINE config 5 size (8 Mi voices) on one GPU, through size-independent
    properties: (1) closed-form phases, state' = state + B*inc for active voices;
    (2) shard linearity: the bus of the whole bank equals the wrapping sum of the buses
    of its 8 contiguous shards (this is exactly the multi-GPU decomposition);
    (3) one shard spot-checked against the oracle."""
    n, shards, nf = 8 << 20, 8, 64
    inc, state = synthetic.saw_bank(n, 0x5EED0005, inc_table, active_fraction=0.95)
    bank = smx.SawBank(n)
    bank.load(inc, state)
    bus, _ = bank.run(nf)
    _, gst = bank.read()
    assert np.array_equal(gst, state + np.uint32(nf) * inc)
    bank.close()
    acc = np.zeros(nf, np.int64)
    per = n // shards
    for s in range(shards):
        sb = smx.SawBank(per)
        sb.load(inc[s * per:(s + 1) * per], state[s * per:(s + 1) * per])
        b, _ = sb.run(nf)
        acc += b
        if s == 3:
            st = state[s * per:(s + 1) * per].copy()
            ob, _ = oracle.synth_run(orc, np.ascontiguousarray(inc[s * per:(s + 1) * per]), st, nf)
            assert np.array_equal(b, ob)
        sb.close()
    assert np.array_equal((acc & 0xFFFFFFFF).astype(np.uint32), bus.view(np.uint32))


def test_rccl_allreduce_path_single_rank(smx, orc, inc_table):
    """The in-library RCCL bus sum (smx_bank_comm_init / allreduce_async / fetch) with a
    1-rank communicator: exercises the second stream, the events and the double-buffered
    bus exactly as the multi-GPU bench does; the sum over one rank is the identity."""
    n = 70000
    inc, state = synthetic.saw_bank(n, 0x5EED0777, inc_table)
    bank = smx.SawBank(n)
    bank.load(inc, state)
    bank.comm_init(0, 1, smx.comm_unique_id())
    st = state.copy()
    for nf in (64, 1, 64, 64, 7):
        bank.run_async(nf)
        bank.allreduce_async(nf)
        want, _ = oracle.synth_run(orc, inc, st, nf)
    bus, vec = bank.fetch(7)
    assert np.array_equal(bus, want)
    # back-to-back steps without fetching in between (the bench's pattern)
    for _ in range(50):
        bank.run_async(64)
        bank.allreduce_async(64)
        want, wvec = oracle.synth_run(orc, inc, st, 64)
    bus, vec = bank.fetch(64)
    assert np.array_equal(bus, want) and np.array_equal(vec.view(np.uint32), wvec.view(np.uint32))
    bank.close()


@pytest.mark.parametrize("frac", [1.0, 0.5, 0.0])
def test_carry_formulation_blocks(smx, orc, inc_table, frac):
    """> 32 frames on >= 2^20 voices run the carry-count formulation (saw_bank.hip):
    full and partial chunks, several chunks per launch, mixed with short blocks that take
    the direct formulation on the same bank."""
    n = (1 << 25) + 1000          # n * frames >= 2^30 selects the carry formulation for >= 64 frames
    inc, state = synthetic.saw_bank(n, 0x5EED0C00, inc_table, active_fraction=frac)
    _check(smx, orc, inc, state, [64, 1, 65, 16, 100, 32, 17])


def test_carry_formulation_extreme_increments(smx, orc):
    """Increments that wrap the phasor every frame (0xFFFFFFFF), never (1), and low-nibble
    patterns of every (phase & 15, inc & 15) class."""
    n = 1 << 25
    k = np.arange(n, dtype=np.uint64)
    inc = np.where(k % 3 == 0, 0xFFFFFFFF, np.where(k % 3 == 1, 1, (k * 2654435761) & 0xFFFFFFFF)).astype(np.uint32)
    state = ((k * 40503 + 12345) & 0xFFFFFFFF).astype(np.uint32)
    _check(smx, orc, inc, state, [64, 130])
    _check(smx, orc, np.full(n, 0x80000000, np.uint32), np.full(n, 0x7FFFFFFF, np.uint32), [64, 64])
    # 2^22 voices x 600 frames: many chunks per launch (MULTI form), 2.5e9 voice-samples
    m = 1 << 22
    _check(smx, orc, np.ascontiguousarray(inc[:m]), np.ascontiguousarray(state[:m]), [600, 513])


def _adversarial_increments(n):
    k = np.arange(n, dtype=np.uint64)
    return np.where(k % 5 == 0, 0xFFFFFFFF, np.where(k % 5 == 1, 1, np.where(k % 5 == 2, 0x80000000,
                    np.where(k % 5 == 3, 0, (k * 2654435761) & 0xFFFFFFFF)))).astype(np.uint32)


_FORMS_ORACLE, _FORMS_ORACLE_SEQ = {}, {}


@pytest.mark.parametrize("form", [1, 2, 0])       # SMX_FORM_STEPPING, SMX_FORM_EVENTS, SMX_FORM_AUTO
def test_long_block_forms_agree(smx, orc, inc_table, form):
    """smx_bank_set_block_form: stepping, wrap events and the device-side automatic choice give the
    oracle's bits on a piano-range bank (where AUTO runs the event form from its first long block on),
    on arbitrary 32-bit increments (many wraps per voice: the event loop runs long, the result must
    not care), on all-wrap / never-wrap / half-scale increments, with voices off, over single
    chunks, partial chunks and multi-chunk launches, and across a reload of the increments.
    A 2^24-voice bank for blocks up to 255 frames (64-frame chunks when stepping; the event form runs 65..255 frames
    as ONE 256-frame chunk that locates wraps only up to the block's last frame), a 2^22-voice bank for 256
    frames and more (256-frame chunks in the event form, 1024-frame chunks from 1024 frames)."""
    for n, piano_blocks, hard_blocks in (((1 << 24) + 2048, [64, 64, 130, 33, 1, 200], [64, 100, 65, 255]),
                                         ((1 << 22) + 1024, [256, 300, 64, 513, 1024, 1030], [256, 257, 1025])):
        inc, state = synthetic.saw_bank(n, 0x5EED0E0E, inc_table, active_fraction=0.9)
        _FORMS_ORACLE_SEQ[n] = []
        bank = smx.SawBank(n)
        bank.set_block_form(form)
        bank.load(inc, state)
        st = state.copy()

        def blocks(frames_list):
            # the three forms are asked for the same blocks of the same banks: the oracle's answer (the full bus of
            # every block and the phases after it) is computed for the first form and kept for the other two
            for nf in frames_list:
                bus, _ = bank.run(nf)
                key = (n, len(_FORMS_ORACLE_SEQ[n]))
                if key not in _FORMS_ORACLE:
                    obus, _ = oracle.synth_run(orc, inc, st, nf)
                    _FORMS_ORACLE[key] = (nf, obus, st.copy())
                knf, obus, kst = _FORMS_ORACLE[key]
                assert knf == nf
                st[:] = kst
                _FORMS_ORACLE_SEQ[n].append(nf)
                assert np.array_equal(bus, obus), "form=%d n=%d frames=%d" % (form, n, nf)

        blocks(piano_blocks)
        assert np.array_equal(bank.read()[1], st)
        inc = _adversarial_increments(n)
        bank.load(inc=inc)                                 # the statistic is computed again: outside the rule, AUTO steps
        blocks(hard_blocks)
        inc = synthetic.saw_bank(n, 0x5EED0E0F, inc_table, active_fraction=1.0)[0]
        bank.load(inc=inc)
        blocks(piano_blocks[:3])
        ginc, gst = bank.read()
        assert np.array_equal(ginc, inc) and np.array_equal(gst, st)
        assert smx.lib().smx_bank_set_block_form(bank._h, 3) == -1
        bank.close()


def test_midi_event_bursts_without_sync(smx, orc):
    """Bursts of MIDI events queued behind each other on the bank's stream (more than the
    4096-slot staging ring between two blocks) must land before the next block's kernel."""
    n = 3000
    bank = smx.SawBank(n)
    n2v = np.zeros(128, np.int32)
    inc = np.zeros(n, np.uint32)
    st = np.zeros(n, np.uint32)
    rng = np.random.default_rng(21)
    for burst in (10, 5000, 300):
        for _ in range(burst):
            msg = np.array([[0x90, 0x80][int(rng.random() < 0.3)], int(rng.integers(0, 128)), int(rng.integers(0, 2)) * 90], np.uint8)
            bank.midi_event(msg)
            orc.orc_midi_event(n2v, inc, n, msg, 3)
        bus, _ = bank.run(64)
        obus, _ = oracle.synth_run(orc, inc, st, 64)
        assert np.array_equal(bus, obus)
    ginc, gst = bank.read()
    assert np.array_equal(ginc, inc) and np.array_equal(gst, st)
    bank.close()


def test_midi_events_batched_equals_event_by_event(smx, orc):
    """smx_bank_midi_events (one copy + one kernel per block) against the oracle's event-by-event
    allocator: bursts larger than the bank (voice 0 stolen over and over), the same voice touched
    many times in one batch, stray note-offs, ignored status bytes, batches growing past the
    initial staging capacity, batched and single-event calls interleaved, and blocks in between
    so that the rebase happens at a non-zero elapsed time."""
    n = 2500
    bank = smx.SawBank(n)
    n2v = np.zeros(128, np.int32)
    inc = np.zeros(n, np.uint32)
    st = np.zeros(n, np.uint32)
    rng = np.random.default_rng(77)
    for burst in (3, 900, 7000, 1, 0, 2000):
        status = rng.choice(np.array([0x90, 0x90, 0x90, 0x80, 0xB0], np.uint8), burst)
        notes = rng.integers(0, 200, burst).astype(np.uint8)          # > 127 exercises note % 128
        vel = (rng.integers(0, 3, burst) * 50).astype(np.uint8)
        msgs = np.stack([status, notes, vel], axis=1) if burst else np.zeros((0, 3), np.uint8)
        bank.midi_events(msgs)
        for m in msgs:
            orc.orc_midi_event(n2v, inc, n, np.ascontiguousarray(m), 3)
        # one more event through the single-event entry, behind the batch
        m = np.array([0x90, int(rng.integers(0, 128)), 100], np.uint8)
        bank.midi_event(m)
        orc.orc_midi_event(n2v, inc, n, m, 3)
        nf = int(rng.integers(1, 100))
        bus, _ = bank.run(nf)
        obus, _ = oracle.synth_run(orc, inc, st, nf)
        assert np.array_equal(bus, obus), burst
    ginc, gst = bank.read()
    assert np.array_equal(ginc, inc) and np.array_equal(gst, st)
    bank.close()


def test_long_blocks_grow_the_bus(smx, orc, inc_table):
    """Blocks longer than the initial 4096-frame bus capacity: the bus buffers (and, for a
    big bank, the carry formulation's scratch slots) are reallocated between blocks."""
    inc, state = synthetic.saw_bank(1500, 0x5EED0B05, inc_table, active_fraction=0.8)
    _check(smx, orc, inc, state, [64, 10000, 7, 4097])
    n = 1 << 20
    inc, state = synthetic.saw_bank(n, 0x5EED0B06, inc_table)
    _check(smx, orc, inc, state, [64, 4500, 64])          # 4500 frames x 2^20 voices: carry path, 71 chunks
    # long launches of smaller banks (from 2^16 voices) reach 2^30 voice-samples too: stepping on the
    # first one, 256-frame event chunks afterwards, short blocks (direct formulation) in between
    for n, frames in (((1 << 18) + 1024, [4096, 64, 4100, 5000]), (1 << 16, [16384, 16390])):
        inc, state = synthetic.saw_bank(n, 0x5EED0B07 + n, inc_table, active_fraction=0.9)
        _check(smx, orc, inc, state, frames)


def test_randomised_block_sequences(smx, orc, inc_table):
    """Seeded fuzz: random bank sizes around every kernel-selection threshold, random block
    lengths, random activity, with MIDI events and bulk reloads between blocks."""
    import os
    # SMX_FUZZ_SEED / SMX_FUZZ_ROUNDS widen the run for a soak (defaults: one round, fixed seed)
    seed = int(os.environ.get("SMX_FUZZ_SEED", "0xF022"), 0)
    rounds = int(os.environ.get("SMX_FUZZ_ROUNDS", "1"))
    rng = np.random.default_rng(seed)
    sizes = [1, 63, 64, 65, 255, 1023, 1024, 1025, 4096, 65535, 65537, (1 << 20) - 1, (1 << 20), (1 << 20) + 1]
    for trial in range(14 * rounds):
        n = sizes[trial % 14]
        inc, state = synthetic.saw_bank(n, 0xF000 + trial + seed, inc_table, active_fraction=float(rng.choice([0.0, 0.3, 1.0])))
        if trial % 3 == 0:                                     # arbitrary (non-table) increments and phases
            inc = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
        bank = smx.SawBank(n)
        bank.load(inc, state)
        n2v = np.zeros(128, np.int32)
        st = state.copy()
        inc = inc.copy()
        for _ in range(6):
            nf = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 257]))
            if n <= 4096 and rng.random() < 0.5:
                msgs = [np.array([0x90, int(rng.integers(0, 128)), int(rng.integers(0, 2)) * 64], np.uint8)
                        for _ in range(int(rng.integers(1, 20)))]
                if rng.random() < 0.5:
                    bank.midi_events(np.stack(msgs))                # the block's events in one call
                else:
                    for msg in msgs:
                        bank.midi_event(msg)
                for msg in msgs:
                    orc.orc_midi_event(n2v, inc, n, msg, 3)
            r = rng.random()
            if r < 0.2:                                        # bulk reload of the increments only
                inc = np.ascontiguousarray(np.roll(inc, 1))
                bank.load(inc=inc)
            elif r < 0.3:                                      # bulk reload of the phases only
                st = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
                bank.load(state=st)
            elif r < 0.4:                                      # read-back in the middle (materialises)
                assert np.array_equal(bank.read()[1], st)
            bus, vec = bank.run(nf)
            obus, ovec = oracle.synth_run(orc, inc, st, nf)
            assert np.array_equal(bus, obus), "n=%d nf=%d" % (n, nf)
            assert np.array_equal(vec.view(np.uint32), ovec.view(np.uint32))
        ginc, gst = bank.read()
        assert np.array_equal(ginc, inc) and np.array_equal(gst, st)
        bank.close()


def test_randomised_long_blocks_big_banks(smx, orc, inc_table):
    """Seeded fuzz over the long-block forms (the sizes of test_randomised_block_sequences stay below
    2^30 voice-samples per launch): banks of 2^22 and 2^24 voices, random block lengths on both sides
    of the 64- and 256-frame chunk sizes, a random form per trial, reloads between piano-range and
    arbitrary increments, voices switched off in between, short blocks mixed in."""
    import os
    seed = int(os.environ.get("SMX_FUZZ_SEED", "0xB16"), 0)
    rng = np.random.default_rng(seed)
    for trial in range(4 * int(os.environ.get("SMX_FUZZ_ROUNDS", "1"))):
        big = trial % 2 == 1
        n = ((1 << 24) + 2048) if big else ((1 << 22) + 1024)
        lengths = [33, 63, 64, 65, 127, 128, 200, 255, 256, 300] if big else [256, 257, 300, 511, 512, 513, 700, 1023, 1024, 1100]
        inc, state = synthetic.saw_bank(n, 0xB160 + trial + seed, inc_table, active_fraction=float(rng.choice([0.5, 1.0])))
        bank = smx.SawBank(n)
        bank.set_block_form(int(rng.integers(0, 3)))
        bank.load(inc, state)
        st = state.copy()
        inc = inc.copy()
        for _ in range(5):
            r = rng.random()
            if r < 0.25:                                       # arbitrary increments
                inc = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
                inc[rng.random(n) < 0.3] = 0
                bank.load(inc=inc)
            elif r < 0.5:                                      # back to a piano-range bank
                inc = synthetic.saw_bank(n, int(rng.integers(1, 1 << 30)), inc_table)[0]
                bank.load(inc=inc)
            elif r < 0.6:
                bank.set_block_form(int(rng.integers(0, 3)))
            nf = int(rng.choice(lengths)) if rng.random() < 0.8 else int(rng.choice([1, 7, 16, 32]))
            bus, _ = bank.run(nf)
            obus, _ = oracle.synth_run(orc, inc, st, nf)
            assert np.array_equal(bus, obus), "trial=%d n=%d nf=%d" % (trial, n, nf)
        ginc, gst = bank.read()
        assert np.array_equal(ginc, inc) and np.array_equal(gst, st)
        bank.close()


def test_pipelined_block_mode(smx, orc, inc_table):
    """SMX_BLOCK_PIPELINED: smx_bank_run returns the previous block (silence first), state and
    note events behave as in sync mode; switching back to sync returns current blocks."""
    n = 5000
    inc, state = synthetic.saw_bank(n, 0x5EED0F1F, inc_table)
    bank = smx.SawBank(n)
    bank.load(inc, state)
    bank.set_block_mode(True)
    st = state.copy()
    want = [np.zeros(64, np.int32)]
    for k in range(6):
        nf = 64 if k != 3 else 32
        bus, vec = bank.run(nf)
        prev = want[-1]
        exp = np.zeros(nf, np.int32)
        m = min(nf, len(prev))
        exp[:m] = prev[:m]
        assert np.array_equal(bus, exp), "block %d" % k
        assert np.array_equal(vec.view(np.uint32), np.array([orc.orc_bus_to_float(int(v)) for v in exp], np.float32).view(np.uint32))
        want.append(oracle.synth_run(orc, inc, st, nf)[0])
    assert np.array_equal(bank.read()[1], st)                # phases are those of the launched blocks
    bank.set_block_mode(False)
    bus, _ = bank.run(64)
    assert np.array_equal(bus, oracle.synth_run(orc, inc, st, 64)[0])
    bank.close()


def test_dropin_tick_functions(smx, orc):
    """sum_tick_saw / sum_tick_square on the caller's struct synth (linux/synth.c:169-195)."""
    L = smx.lib()
    x = smx.Synth()
    L.synth_init(C.byref(x))
    inc = np.zeros(64, np.uint32)
    st = np.zeros(64, np.uint32)
    n2v = np.zeros(128, np.int32)
    for note in (40, 52, 59, 64, 100):
        L.synth_note_on(C.byref(x), note)
        orc.orc_note_on(n2v, inc, 64, note)
    for i in range(40):
        if i % 2:
            got, want = L.sum_tick_square(C.byref(x)), orc.orc_sum_tick_square(inc, st, 64)
        else:
            got, want = L.sum_tick_saw(C.byref(x)), orc.orc_bus_to_float(orc.orc_sum_tick_saw(inc, st, 64))
        assert np.float32(got).view(np.uint32) == np.float32(want).view(np.uint32)
    assert [x.voice[v].note_state for v in range(64)] == st.tolist()


def test_tick_kernel_1_to_4_frames(smx, orc, inc_table):
    """<= 4-frame blocks of 2^20..2^26-voice banks (n a multiple of 4096) run saw_tick_kernel
    (1024-thread workgroups): every frame count 1..4, off voices, repeated ticks."""
    n = 1 << 21
    inc, state = synthetic.saw_bank(n, 0x5EED0A11, inc_table, active_fraction=0.85)
    _check(smx, orc, inc, state, [1, 2, 3, 4, 1, 1, 64, 3])


def test_billion_voice_bank_index_safety(smx):
    """2^30 + 3072 voices (8 GiB of bank state; every byte offset beyond 2^32): the 1-frame tick
    path, a 16-frame block (direct formulation with bus slots) and a 64-frame block (carry
    formulation) against the closed form  bus[t] = sum_v ((int32)(state_v + t*inc_v) >> 4)
    evaluated with numpy for frames 0, 1, 15 and 63, then the phases read back; then the event forms of 32, 64 and 128
    frames, pinned.  Voice values depend on the index, so a workgroup reading the wrong rows (32-bit index wrap) changes
    the sums."""
    n = (1 << 30) + 3072
    idx = np.arange(n, dtype=np.uint64)
    inc = ((idx * np.uint64(2654435761)) >> np.uint64(7)).astype(np.uint32) | np.uint32(1)
    state = ((idx * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(29)).astype(np.uint32)
    inc[5::1000003] = 0                                   # a few voices off
    del idx

    def closed_form(t):
        ph = state + np.uint32(t) * inc
        ph[inc == 0] = 0                                  # off voices contribute nothing
        return np.int64((ph.view(np.int32) >> 4).sum(dtype=np.int64))

    def wrap(x):
        return np.int32(np.uint32(x & 0xFFFFFFFF))

    bank = smx.SawBank(n)
    bank.load(inc, state)
    f0, f1, f15, f63 = (closed_form(t) for t in (0, 1, 15, 63))
    # off voices are parked at phase 0 and do not advance: their closed-form term is 0 at every t
    bus, _ = bank.run(1)
    assert bus[0] == wrap(f0)
    bank.load(inc, state)
    bus, _ = bank.run(16)
    assert bus[0] == wrap(f0) and bus[1] == wrap(f1) and bus[15] == wrap(f15)
    bank.load(inc, state)
    bus, _ = bank.run(64)
    assert bus[0] == wrap(f0) and bus[1] == wrap(f1) and bus[15] == wrap(f15) and bus[63] == wrap(f63)
    ginc, gst = bank.read()
    assert np.array_equal(ginc, inc)
    want = state + np.uint32(64) * inc                    # inc == 0: unchanged
    assert np.array_equal(gst, want)
    # the event forms pinned (arbitrary increments: ~32 wraps per voice and 64 frames, the loops run long): the 64-frame
    # kernel in 512-thread workgroups, the 32-frame chunk, the 128-frame chunk in 1024-thread workgroups
    bank.set_block_form(2)
    f127 = closed_form(127)
    for nf, checks in ((64, ((0, f0), (1, f1), (15, f15), (63, f63))), (32, ((0, f0), (1, f1), (15, f15))),
                       (128, ((0, f0), (63, f63), (127, f127)))):
        bank.load(inc, state)
        bus, _ = bank.run(nf)
        for t, f in checks:
            assert bus[t] == wrap(f), ("events", nf, t)
    bank.close()


@pytest.mark.parametrize("n", [4097, 8192, 8193, 12288, 65536 + 1024, (1 << 18) + 5])
def test_small_bank_chunk_edges(smx, orc, inc_table, n):
    """Banks below 2^20 voices cut blocks into 16-frame chunks on blockIdx.y (1 voice per lane below 2^13
    padded voices, 4 from there on): every chunk boundary, ragged last chunks, and the lazily materialised phase
    (tbase != 0) across them."""
    inc, state = synthetic.saw_bank(n, 0x5EED0600 + n, inc_table, active_fraction=0.85)
    _check(smx, orc, inc, state, [15, 16, 17, 1, 31, 32, 33, 47, 48, 49, 63, 64, 65, 100, 5, 257])


def test_auto_form_statistic_is_conservative_under_note_events(smx, orc, inc_table):
    """AUTO picks the wrap-event form only while the bank's increments are inside the rule (largest below
    6.5 * 2^26, mean at most 2^27).  The pick is exact as of the last long block and every note event in between
    keeps it conservative: one voice above the bound, or a running sum above the bound, selects stepping AT ONCE
    (before any further block), and the next long block recomputes it exactly.  Bits are the same throughout."""
    FORM_STEPPING, FORM_EVENTS = 1, 2
    n = 1 << 24
    inc, state = synthetic.saw_bank(n, 0x5EED0700, inc_table)          # piano range: inside the rule
    inc[:4096] = 0                                                      # free voices for the note-ons below
    bank = smx.SawBank(n)
    bank.load(inc, state)
    st = state.copy()

    def block():
        bus, _ = bank.run(64)
        obus, _ = oracle.synth_run(orc, inc, st, 64, want_vec=False)
        assert np.array_equal(bus, obus)

    assert bank.next_block_form() == FORM_EVENTS                        # after a load: the rule on the loaded increments (round 3)
    block()
    assert bank.next_block_form() == FORM_EVENTS                        # measured again by that block: inside the rule
    block()
    # one very high voice (MIDI 127: 16.7 wraps per 64 frames): stepping at once
    n2v = np.zeros(128, np.int32)
    bank.note_on(127)
    orc.orc_note_on(n2v, inc, n, 127)
    assert bank.next_block_form() == FORM_STEPPING
    block()
    assert bank.next_block_form() == FORM_STEPPING                      # exact statistic: the voice is still there
    bank.note_off(127)
    orc.orc_note_off(n2v, inc, n, 127)
    assert bank.next_block_form() == FORM_STEPPING                      # conservative until the next long block
    block()
    assert bank.next_block_form() == FORM_EVENTS
    # a burst of note-ons below the per-voice bound that lifts the MEAN above 2 wraps: the running sum trips it
    # (the bank's mean is ~1.1 wraps; 4096 voices cannot do it, so load a bank close to the bound first)
    inc2 = np.full(n, (1 << 27) - 3000, np.uint32)
    inc2[:4096] = 0
    inc[:] = inc2
    bank.load(inc=inc2)
    assert bank.next_block_form() == FORM_EVENTS                        # mean just below 2 wraps: known from the load on
    block()
    assert bank.next_block_form() == FORM_EVENTS
    hi = inc2.copy()
    hi[5000] = 15 << 25                                                 # one voice above the per-voice bound
    bank.load(inc=hi)
    assert bank.next_block_form() == FORM_STEPPING                      # ... and so is this
    bank.load(inc=inc2)
    ev = np.array([[0x90, 100 + (k % 8), 100] for k in range(3000)], np.uint8)   # notes 100..107: 3.6 .. 5.3 wraps each
    bank.midi_events(ev)
    for m in ev:
        orc.orc_midi_event(n2v, inc, n, np.ascontiguousarray(m), 3)
    assert bank.next_block_form() == FORM_STEPPING
    block()
    block()
    _, gst = bank.read()
    assert np.array_equal(gst, st)
    bank.close()


def test_deferred_slot_fold_sequences(smx, orc, inc_table):
    """Blocks of 5+ frames on >= 2^20 voices below the carry crossover leave the fold of their slots to the NEXT
    launch (smx_common.h SawPending) or to whoever reads the bus.  Un-fetched blocks in a row, every kind of
    successor (slot launch with another chunk length, tick kernel, carry forms, square variant, a longer block that
    replaces the ring), events and reloads in between: each fetched bus must equal the oracle's."""
    rng = np.random.default_rng(0xF01D ^ int(os.environ.get("SMX_FUZZ_SEED", "0"), 0))
    rounds = int(os.environ.get("SMX_FUZZ_ROUNDS", "1"))
    n = (1 << 20) + 5
    inc, state = synthetic.saw_bank(n, 0x5EED0F01, inc_table, active_fraction=0.8)
    bank = smx.SawBank(n)
    bank.load(inc, state)
    st = state.copy()
    frames = [5, 8, 9, 16, 17, 32, 33, 64, 65, 100, 128, 200, 1, 3, 4]
    n2v = np.zeros(128, np.int32)
    inc = inc.copy()
    checked = 0
    for step in range(120 * rounds):
        nf = int(rng.choice(frames)) if step != 60 else 5000          # 5000: the ring and the scratch are replaced
        r = rng.random()
        if r < 0.08:
            got = bank.run_square(nf)
            want = np.array([orc.orc_sum_tick_square(inc, st, n) for _ in range(nf)], np.float32)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), ("square", step, nf)
            checked += 1
            continue
        if r < 0.16:
            note = int(rng.integers(0, 128))
            bank.note_on(note)
            orc.orc_note_on(n2v, inc, n, note)
        bank.run_async(nf)
        want, wvec = oracle.synth_run(orc, inc, st, nf)
        if rng.random() < 0.4 or step in (59, 60, 61, 120 * rounds - 1):
            bus, vec = bank.fetch(nf)
            assert np.array_equal(bus, want), ("bus", step, nf)
            assert np.array_equal(vec.view(np.uint32), wvec.view(np.uint32))
            checked += 1
    assert checked >= 40
    ginc, gst = bank.read()
    assert np.array_equal(ginc, inc) and np.array_equal(gst, st)
    bank.close()
    # the same through a communicator (1 rank): groups of 8 un-fetched blocks per all-reduce
    bank = smx.SawBank(n)
    bank.load(inc, state)
    bank.comm_init(0, 1, smx.comm_unique_id())
    st = state.copy()
    for step in range(40):
        nf = int(rng.choice([8, 16, 64, 3, 100]))
        bank.run_async(nf)
        bank.allreduce_async(nf)
        want, _ = oracle.synth_run(orc, inc, st, nf)
        if step % 7 == 6 or step == 39:
            assert np.array_equal(bank.fetch(nf)[0], want), ("comm", step, nf)
    bank.close()


def _bus_at(inc, st, frames):
    """sum_tick_saw at the given frames of a block that starts with phases st (numpy; the phasor is linear)."""
    on = inc != 0
    out = []
    with np.errstate(over="ignore"):
        for f in frames:
            ph = st + np.uint32(f) * inc
            out.append(int(np.where(on, ph.view(np.int32) >> 4, 0).sum(dtype=np.int64)))
    return ((np.array(out, np.int64) + (1 << 31)) % (1 << 32) - (1 << 31)).astype(np.int32)


@pytest.mark.parametrize("form", [1, 2, 0])
def test_long_block_sequences_with_a_pinned_form(smx, orc, inc_table, form):
    """Long blocks of a 2^24-voice bank (the carry formulations) with a form pinned by the caller (1 stepping, 2 events)
    or, under AUTO (0), by the host after four equal picks: un-fetched long blocks in a row (one chunk and several),
    every kind of successor (another long block of either chunk count, a 256+-frame launch with its own slot layout,
    direct slot launches, the tick kernel, note events, a reload, the square variant): every fetched bus (three frames
    of it, against the closed form of the linear phasor: the bank is too big to step on the CPU in test time) and
    the final phases are right, and AUTO's pick stays conservative after a high note.  (Written for round 3's
    attempt to defer the carry forms' finalize to the next launch -- measured slower and reverted; the
    sequences stay as a test of the long-block path.)"""
    rng = np.random.default_rng(0xCA77 + form)
    n = 1 << 24                                   # 2^24 voices x 64 frames = 2^30 voice-samples: the carry path
    inc, state = synthetic.saw_bank(n, 0x5EED0F02, inc_table, active_fraction=0.9)
    bank = smx.SawBank(n)
    bank.load(inc, state)
    bank.set_block_form(form)
    st = state.copy()
    inc = inc.copy()
    n2v = np.zeros(128, np.int32)
    frames = [64, 64, 64, 128, 100, 65, 256, 300, 16, 8, 1, 3, 33]
    checked = 0
    for step in range(48):
        nf = int(rng.choice(frames)) if step >= 8 else 64         # a run of long blocks first (AUTO: the host pins)
        r = rng.random()
        if step >= 8 and r < 0.10:
            note = int(rng.integers(0, 128))
            bank.note_on(note)
            orc.orc_note_on(n2v, inc, n, note)
        elif step >= 8 and r < 0.14:
            bank.load(inc=inc)                                       # same increments: a reload between long blocks
        elif step >= 8 and r < 0.18:
            got = bank.run_square(2)
            want = np.array([orc.orc_sum_tick_square(inc, st, n) for _ in range(2)], np.float32)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), ("square", step)
            continue
        bank.run_async(nf)
        if rng.random() < 0.3 or step in (7, 47):
            pick = sorted({0, nf // 2, nf - 1})
            bus, _ = bank.fetch(nf)
            assert np.array_equal(bus[pick], _bus_at(inc, st, pick)), ("bus", form, step, nf)
            checked += 1
        with np.errstate(over="ignore"):
            st += np.uint32(nf) * inc                                # an off voice (inc 0) does not advance
    assert checked >= 8
    if form == 0:
        # a note far above the rule's bound: the very next long block must step, whatever finalizes were in flight
        bank.note_on(127)
        orc.orc_note_on(n2v, inc, n, 127)
        assert bank.next_block_form() == 1
        bank.run_async(64)
        assert np.array_equal(bank.fetch(64)[0][[0, 63]], _bus_at(inc, st, [0, 63]))
        with np.errstate(over="ignore"):
            st += np.uint32(64) * inc
    ginc, gst = bank.read()
    assert np.array_equal(ginc, inc) and np.array_equal(gst, st)
    bank.close()


@pytest.mark.parametrize("n", [64, 5000, 1 << 17, (1 << 20) + 4096, 1 << 23])
def test_load_run_replaces_both_arrays_on_any_bank(smx, orc, inc_table, n):
    """smx_bank_load_run (new increments + phases + one block, one synchronisation) on banks of every launch class:
    long blocks of >= 2^23-voice banks (the carry formulations) take the bank's sum of increments from the scratch
    header, which the call has to bring up to date like smx_bank_load does; between the calls ordinary blocks, an un-fetched block whose
    slot fold is still owed, and a bank that had pinned a form."""
    bank = smx.SawBank(n)
    rng = np.random.default_rng(n)
    inc0, st0 = synthetic.saw_bank(n, 0x5EED0A00, inc_table, active_fraction=0.8)
    bank.load(inc0, st0)
    for _ in range(6):
        bank.run_async(64)                                     # AUTO may pin a form on the old increments
    st = None
    for k, nf in enumerate([64, 1, 128, 64, 33, 256, 64]):
        inc, st = synthetic.saw_bank(n, 0x5EED0A01 + k, inc_table, active_fraction=0.3 + 0.1 * k)
        if k == 3:
            inc = np.where(rng.random(n) < 0.5, inc, np.uint32(0xF0000001)).astype(np.uint32)   # far above the event form's bound
        bus, vec = bank.load_run(inc, st, nf)
        pick = sorted({0, nf // 2, nf - 1})
        assert np.array_equal(bus[pick], _bus_at(inc, st, pick)), (n, k, nf)
        if n <= (1 << 17):
            obus, ovec = oracle.synth_run(orc, inc, st.copy(), nf)
            assert np.array_equal(bus, obus) and np.array_equal(vec.view(np.uint32), ovec.view(np.uint32))
        with np.errstate(over="ignore"):
            st = st + np.uint32(nf) * inc
        nf2 = [64, 16, 1][k % 3]
        bank.run_async(nf2)                                    # left un-fetched: the next load_run meets its owed fold
        with np.errstate(over="ignore"):
            st = st + np.uint32(nf2) * inc
    ginc, gst = bank.read()
    assert np.array_equal(ginc, inc) and np.array_equal(gst, st)
    bank.close()


@pytest.mark.parametrize("form", [0, 2, 1])
def test_short_chunk_forms_17_to_32_frames(smx, orc, inc_table, form):
    """Blocks of 17..32 frames of a >= 2^25-voice bank run as ONE 32-frame chunk of the carry formulations (AUTO: the
    device-side flag picks stepping or located wraps; EVENTS pinned; STEPPING pinned keeps the direct form): a ragged
    bank with off voices, every frame count of the class next to its neighbours (16, 33, 64) and the tick kernel,
    un-fetched blocks in between, a note far above the event form's bound (AUTO must step from the next block on), a
    reload; three frames of every fetched bus against the closed form of the linear phasor, and the final phases."""
    n = (1 << 25) + 4096 + 7
    rng = np.random.default_rng(0x5C17 + form)
    inc, state = synthetic.saw_bank(n, 0x5EED0C17, inc_table, active_fraction=0.9)
    bank = smx.SawBank(n)
    bank.load(inc, state)
    bank.set_block_form(form)
    st, inc = state.copy(), inc.copy()
    n2v = np.zeros(128, np.int32)
    frames = [32, 32, 32, 32, 32, 32, 17, 24, 31, 18, 16, 33, 64, 32, 1, 25, 32, 32]
    for step, nf in enumerate(frames + [int(x) for x in rng.integers(15, 35, 10)]):
        if step == 12:
            if form == 0:
                assert bank.next_block_form() == 2                 # a piano-range bank: inside the rule
            bank.note_on(127)                                      # 13.3 wraps per 64 frames: above the bound
            orc.orc_note_on(n2v, inc, n, 127)
            if form == 0:
                assert bank.next_block_form() == 1
        if step == 20:
            inc = synthetic.saw_bank(n, 0x5EED0C18, inc_table, active_fraction=0.8)[0]
            bank.load(inc=inc)
        bank.run_async(nf)
        if step % 3 != 1:
            pick = sorted({0, nf // 2, nf - 1})
            assert np.array_equal(bank.fetch(nf)[0][pick], _bus_at(inc, st, pick)), (form, step, nf)
        with np.errstate(over="ignore"):
            st += np.uint32(nf) * inc
    ginc, gst = bank.read()
    assert np.array_equal(ginc, inc) and np.array_equal(gst, st)
    bank.close()
