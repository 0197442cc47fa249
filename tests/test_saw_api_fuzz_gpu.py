"""GPU: model-based fuzz of the saw bank's WHOLE C-ABI surface (linux/synth.c:145-206 over N voices).  Random sequences
of every call that changes or reads the bank -- load (both / increments / phases), load_run, note on / off one by
one and as a block's batch, synchronous, un-fetched and fetched asynchronous blocks of every length class, the square
variant, read-back, sync, form and block-mode switches -- on banks of every launch class, against a host model of
the same calls: the CPU oracle where the bank is small, the closed form of the linear phasor (three frames per
block) where stepping 2^23 voices on the CPU would take too long.  What the sequences are after is state that one
call leaves for another: lazy phases, owed folds, the scratch header's statistics, the pinned form, the published bus.
SMX_FUZZ_SEED / SMX_FUZZ_ROUNDS widen the run for a soak."""
import os

import numpy as np
import pytest

import oracle
from synth_tools_amd import synthetic

pytestmark = pytest.mark.gpu


def _bus_at(inc, st, frames):
    on = inc != 0
    out = []
    with np.errstate(over="ignore"):
        for f in frames:
            ph = st + np.uint32(f) * inc
            out.append(int(np.where(on, ph.view(np.int32) >> 4, 0).sum(dtype=np.int64)))
    return ((np.array(out, np.int64) + (1 << 31)) % (1 << 32) - (1 << 31)).astype(np.int32)


class Model:
    """The bank as the reference would hold it: increments, phases, note2voice."""

    def __init__(self, orc, n, inc, st):
        self.orc, self.n = orc, n
        self.inc, self.st = inc.copy(), st.copy()
        self.n2v = np.zeros(128, np.int32)

    def block(self, nf):
        """-> (frames checked, their bus); advances the phases"""
        if self.n <= (1 << 16):
            bus, _ = oracle.synth_run(self.orc, self.inc, self.st, nf)          # advances self.st
            return np.arange(nf), bus
        pick = np.array(sorted({0, nf // 2, nf - 1}))
        bus = _bus_at(self.inc, self.st, pick)
        with np.errstate(over="ignore"):
            self.st += np.uint32(nf) * self.inc                                 # an off voice (inc 0) stays
        return pick, bus

    def square(self, k):
        return np.array([self.orc.orc_sum_tick_square(self.inc, self.st, self.n) for _ in range(k)], np.float32)

    def midi(self, msg):
        self.orc.orc_midi_event(self.n2v, self.inc, self.n, msg, 3)


@pytest.mark.parametrize("n", [64, 5000, 1 << 16, (1 << 20) + 4096, 1 << 23])
def test_every_call_in_random_order(smx, orc, inc_table, n):
    seed = int(os.environ.get("SMX_FUZZ_SEED", "0xA91"), 0)
    rounds = int(os.environ.get("SMX_FUZZ_ROUNDS", "1"))
    rng = np.random.default_rng(seed + n)
    lengths = [1, 2, 4, 5, 16, 17, 32, 33, 64, 64, 64, 65, 128, 256, 300]
    for trial in range(2 * rounds):
        inc, st = synthetic.saw_bank(n, 0xA910 + trial + seed, inc_table, active_fraction=float(rng.choice([0.4, 0.95])))
        bank = smx.SawBank(n)
        bank.load(inc, st)
        m = Model(orc, n, inc, st)
        pipelined = False
        pipe_prev = None                     # (frames, bus) the next pipelined run hands out; None: silence
        log = []
        steps = 60 if n <= (1 << 20) + 4096 else 40
        for step in range(steps):
            r = rng.random()
            nf = int(rng.choice(lengths))
            try:
                if r < 0.05:
                    m.inc, m.st = synthetic.saw_bank(n, int(rng.integers(1, 1 << 30)), inc_table,
                                                     active_fraction=float(rng.choice([0.2, 0.9])))
                    bank.load(m.inc, m.st); log.append("load both")
                elif r < 0.09:
                    if rng.random() < 0.5:
                        m.inc = synthetic.saw_bank(n, int(rng.integers(1, 1 << 30)), inc_table)[0]
                    else:                                               # arbitrary increments, far above the event form's bound
                        m.inc = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
                        m.inc[rng.random(n) < 0.3] = 0
                    bank.load(inc=m.inc); log.append("load inc")
                elif r < 0.12:
                    m.st = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
                    bank.load(state=m.st); log.append("load state")
                elif r < 0.17 and not pipelined:
                    m.inc, m.st = synthetic.saw_bank(n, int(rng.integers(1, 1 << 30)), inc_table,
                                                     active_fraction=float(rng.choice([0.3, 1.0])))
                    bus, vec = bank.load_run(m.inc, m.st, nf); log.append("load_run %d" % nf)
                    pick, want = m.block(nf)
                    assert np.array_equal(bus[pick], want)
                elif r < 0.27:
                    msgs = np.stack([np.array([0x90 if rng.random() < 0.8 else 0x80, int(rng.integers(0, 128)),
                                               int(rng.integers(0, 2)) * 64], np.uint8)
                                     for _ in range(int(rng.integers(1, 12)))])
                    if rng.random() < 0.5:
                        bank.midi_events(msgs); log.append("midi batch %d" % len(msgs))
                    else:
                        for msg in msgs:
                            bank.midi_event(msg)
                        log.append("midi x%d" % len(msgs))
                    for msg in msgs:
                        m.midi(msg)
                elif r < 0.30:
                    note = int(rng.integers(0, 128))
                    if rng.random() < 0.6:
                        bank.note_on(note); orc.orc_note_on(m.n2v, m.inc, n, note); log.append("on %d" % note)
                    else:
                        bank.note_off(note); orc.orc_note_off(m.n2v, m.inc, n, note); log.append("off %d" % note)
                elif r < 0.34:
                    ginc, gst = bank.read(); log.append("read")
                    assert np.array_equal(ginc, m.inc) and np.array_equal(gst, m.st)
                elif r < 0.37:
                    f = int(rng.integers(0, 3)); bank.set_block_form(f); log.append("form %d" % f)
                elif r < 0.40:
                    pipelined = not pipelined
                    bank.set_block_mode(pipelined); log.append("pipelined %d" % pipelined)
                    pipe_prev = None
                elif r < 0.43:
                    bank.sync(); log.append("sync")
                elif r < 0.46 and not pipelined:
                    k = int(rng.integers(1, 4))
                    got = bank.run_square(k); log.append("square %d" % k)
                    assert np.array_equal(got.view(np.uint32), m.square(k).view(np.uint32))
                elif r < 0.70 and not pipelined:
                    bank.run_async(nf); log.append("async %d" % nf)
                    pick, want = m.block(nf)
                    if rng.random() < 0.5:
                        bus, vec = bank.fetch(nf); log.append("fetch")
                        assert np.array_equal(bus[pick], want)
                        fl = np.array([orc.orc_bus_to_float(int(v)) for v in want], np.float32)
                        assert np.array_equal(vec[pick].view(np.uint32), fl.view(np.uint32))
                else:
                    bus, vec = bank.run(nf); log.append("run %d" % nf)
                    pick, want = m.block(nf)
                    if pipelined:
                        exp = np.zeros(nf, np.int64)
                        mask = np.zeros(nf, bool)                        # frames of this answer the model knows
                        if pipe_prev is None:
                            mask[:] = True                               # the first call returns silence
                        else:
                            pf, pb, plen = pipe_prev
                            mask[plen:] = True                           # beyond the previous block: zeros
                            ok = pf < nf
                            exp[pf[ok]] = pb[ok]
                            mask[pf[ok]] = True
                        assert np.array_equal(bus[mask], exp[mask].astype(np.int32))
                        pipe_prev = (pick, want, nf)
                    else:
                        assert np.array_equal(bus[pick], want)
            except AssertionError:
                raise AssertionError("n=%d trial=%d step=%d after: %s" % (n, trial, step, " | ".join(log[-12:])))
        if pipelined:
            bank.set_block_mode(False)
        ginc, gst = bank.read()
        assert np.array_equal(ginc, m.inc) and np.array_equal(gst, m.st), " | ".join(log[-12:])
        bank.close()


def test_banks_on_concurrent_host_threads(smx, orc, inc_table):
    """One bank per host thread, four threads at once (ctypes releases the GIL inside every call): each bank has its own
    stream, pinned words and error string; nothing in the library is shared between banks except lazily initialised
    read-only switches.  Every thread checks its blocks against the oracle (a second oracle call runs under the GIL:
    the product calls are what overlap)."""
    from concurrent.futures import ThreadPoolExecutor

    def worker(k):
        n = [64, 5000, 70000, (1 << 20) + 4096][k]
        rng = np.random.default_rng(0x7123 + k)
        inc, st = synthetic.saw_bank(n, 0x5EED0710 + k, inc_table, active_fraction=0.8)
        bank = smx.SawBank(n)
        bank.load(inc, st)
        st = st.copy()
        pdm = smx.PdmBank(3000 + k)
        sp, ac = synthetic.pdm_bank(3000 + k, 0x710 + k)
        pdm.load(sp, ac)
        oa = ac.copy()
        for step in range(25):
            nf = int(rng.choice([1, 16, 33, 64, 64, 128]))
            if step % 4 == 3:
                bank.run_async(nf)
                bus = bank.fetch(nf)[0]
            else:
                bus = bank.run(nf)[0]
            want, _ = oracle.synth_run(orc, inc, st, nf)
            assert np.array_equal(bus, want), (k, step, nf)
            nt = int(rng.choice([1, 2, 40]))
            assert np.array_equal(pdm.tick_n(nt), oracle.pdm_run(orc, sp, oa, nt)), (k, step, "pdm")
        assert np.array_equal(bank.read()[1], st)
        bank.close()
        pdm.close()
        return k

    with ThreadPoolExecutor(max_workers=4) as ex:
        assert sorted(ex.map(worker, range(4))) == [0, 1, 2, 3]
