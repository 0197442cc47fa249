"""One rank of a multi-process run of the PRODUCT's sharded saw bank (test helper, started by
tests/test_multi_rank_gpu.py): its shard on the device it is given, the library's own RCCL
communicator (unique id over synth_tools_amd.rendezvous), every mode that returns a reduced bus,
each compared with the oracle run over ALL shards.  Prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def allocator_mode(rank, world, per, bank, orc, tab, oracle, synthetic):
    """The reference's allocator (linux/synth.c:145-165) over the GLOBAL bank, replicated on every rank
    (smx_bank_shard): the same MIDI stream on every rank; compared with the oracle's allocator over one array of
    world * per voices -- the reduced bus of every block, and at the end this rank's slice of inc[] / state[]."""
    total = world * per
    bank.load(np.zeros(per, np.uint32), np.zeros(per, np.uint32))     # a fresh bank (main() had loaded a shard)
    bank.shard(rank * per, total)
    rng = np.random.default_rng(4711)                       # the SAME stream on every rank
    g_inc = np.zeros(total, np.uint32)
    g_st = np.zeros(total, np.uint32)
    n2v = np.zeros(128, np.int32)
    checks = 0

    def block(nf):
        bus, _ = bank.run(nf)
        obus, _ = oracle.synth_run(orc, g_inc, g_st, nf, want_vec=False)
        assert np.array_equal(bus, obus), "bus"

    # (a) a fresh bank: note-ons fill the global bank in voice order, across the shard boundaries
    for k in range(40):
        note = int(rng.integers(0, 128))
        if rng.random() < 0.7:
            bank.note_on(note); orc.orc_note_on(n2v, g_inc, total, note)
        else:
            bank.note_off(note); orc.orc_note_off(n2v, g_inc, total, note)       # incl. stray note-offs -> voice 0
        if k % 5 == 4:
            block(64); checks += 1
    # (b) a loaded bank (collective load: the ranks exchange their free maps), a few free voices in every shard
    g_inc[:], g_st[:] = synthetic.saw_bank(total, 0x5EED0A11, tab, active_fraction=0.97)
    n2v[:] = 0
    bank.shard(rank * per, total)                           # resets the note table, as a fresh synth_init would
    bank.load(g_inc[rank * per:(rank + 1) * per], g_st[rank * per:(rank + 1) * per])
    block(64); checks += 1
    # batches of events (one copy + one kernel per block on the owner ranks) until the bank is full, then beyond:
    # the allocator steals global voice 0 (rank 0's) on every further note-on
    saw_full = False
    for blk in range(12):
        ev = np.zeros((40, 3), np.uint8)
        ev[:, 0] = np.where(rng.random(40) < 0.75, 0x90, 0x80)
        ev[:, 1] = rng.integers(0, 128, 40)
        ev[:, 2] = np.where(rng.random(40) < 0.9, 100, 0)    # velocity 0 = note-off
        bank.midi_events(ev)
        for m in ev:
            saw_full = saw_full or not np.any(g_inc == 0)    # a note-on into a full bank steals global voice 0
            orc.orc_midi_event(n2v, g_inc, total, np.ascontiguousarray(m), 3)
        block(int(rng.choice([1, 16, 64]))); checks += 1
    full = saw_full
    inc, st = bank.read()
    assert np.array_equal(inc, g_inc[rank * per:(rank + 1) * per]), "inc slice"
    assert np.array_equal(st, g_st[rank * per:(rank + 1) * per]), "state slice"
    print(json.dumps({"rank": rank, "ranks_seen": bank.comm_ranks(), "checks": checks, "bank_filled_up": bool(full)}))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    device = int(os.environ.get("SMX_TEST_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    per = int(os.environ.get("SMX_TEST_VOICES", "5000"))
    import oracle
    import synth_tools_amd as sta
    from synth_tools_amd import rendezvous, synthetic
    orc = oracle.load()
    tab = synthetic.note_inc_table(sta.lib().note_to_inc)
    shards = [synthetic.saw_bank(per, 0x5EED0500 + r, tab, active_fraction=0.9) for r in range(world)]
    all_inc = np.concatenate([s[0] for s in shards])
    all_st = np.concatenate([s[1] for s in shards])
    rdzv = rendezvous.Rendezvous(rank, world, timeout=60)
    bank = sta.SawBank(per, device=device)
    bank.load(*shards[rank])
    uid = rdzv.broadcast(sta.comm_unique_id().tobytes() if rank == 0 else b"")
    bank.comm_init(rank, world, np.frombuffer(uid, np.uint8).copy())
    res = {"rank": rank, "ranks_seen": bank.comm_ranks(), "checks": 0}

    def expect(nf):
        return oracle.synth_run(orc, all_inc, all_st, nf)

    if os.environ.get("SMX_TEST_ALLOCATOR"):
        allocator_mode(rank, world, per, bank, orc, tab, oracle, synthetic)
        rdzv.barrier()
        bank.close()
        rdzv.close()
        return
    if os.environ.get("SMX_TEST_VIOLATE"):
        # the SPMD contract broken on purpose: only rank 0 fetches in the middle of a group, so the ranks would
        # issue different collectives (3 blocks + 5 blocks vs 8 blocks).  With the test double this aborts.
        for k in range(8):
            bank.run_async(64)
            bank.allreduce_async(64)
            if k == 2 and rank == 0:
                bank.fetch(64)
        bank.sync()
        print(json.dumps({"rank": rank, "violation_went_unnoticed": True}))
        return
    # (1) sync mode: smx_bank_run returns the sum over all ranks
    for nf in (64, 1, 7, 64, 300):
        bus, vec = bank.run(nf)
        obus, ovec = expect(nf)
        assert np.array_equal(bus, obus), ("sync", nf)
        assert np.array_equal(vec.view(np.uint32), ovec.view(np.uint32))
        res["checks"] += 1
    # (2) grouped: 20 blocks in flight, one collective per 8, the last one fetched
    before = bank.comm_stats()
    want = None
    for k in range(20):
        bank.run_async(64)
        bank.allreduce_async(64)
        want = expect(64)[0]
    bus, _ = bank.fetch(64)
    assert np.array_equal(bus, want), "grouped"
    after = bank.comm_stats()
    res["grouped_collectives"] = after[0] - before[0]
    res["grouped_block_sums"] = after[1] - before[1]
    assert res["grouped_block_sums"] == 20 and res["grouped_collectives"] <= 5
    res["checks"] += 1
    # (3) every block of a grouped run is reduced (fetch each one right after its request)
    bank.set_comm_group(4)
    for k in range(9):
        bank.run_async(16)
        bank.allreduce_async(16)
        bus, _ = bank.fetch(16)
        assert np.array_equal(bus, expect(16)[0]), ("grouped-each", k)
        res["checks"] += 1
    bank.set_comm_group(8)
    # (4) pipelined mode: block k-1 of the SUM comes back while block k runs
    bank.sync()
    bank.set_block_mode(1)
    prev = np.zeros(64, np.int32)
    for k in range(6):
        bus, _ = bank.run(64)
        assert np.array_equal(bus, prev), ("pipelined", k)
        prev = expect(64)[0]
        res["checks"] += 1
    bank.set_block_mode(0)
    bank.sync()
    # (5) sum_tick_square (linux/synth.c:182-195): the OR over every rank's voices
    for nf in (64, 5):
        got = bank.run_square(nf)
        want = np.array([orc.orc_sum_tick_square(all_inc, all_st, len(all_inc)) for _ in range(nf)], np.float32)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "square"
        res["checks"] += 1
    # (6) synchronous blocks between asynchronous ones: the synchronous block's sum is issued on the COMPUTE stream
    # (round 3), after the sums that earlier asynchronous blocks still have queued and after whatever the comm stream
    # has in flight; asynchronous groups after it go back to the comm stream
    for rnd in range(3):
        wants = []
        for k in range(3 + rnd):                       # queued, not yet issued (group of 8)
            bank.run_async(64)
            bank.allreduce_async(64)
            wants.append(expect(64)[0])
        bus, _ = bank.run(17 + rnd)                    # synchronous: flushes the queue first, then its own sum
        assert np.array_equal(bus, expect(17 + rnd)[0]), ("sync after queued", rnd)
        for k in range(9):                             # a whole group + one: the comm stream again
            bank.run_async(64)
            bank.allreduce_async(64)
            want = expect(64)[0]
        assert np.array_equal(bank.fetch(64)[0], want), ("async after sync", rnd)
        res["checks"] += 2
    # (7) a seeded random sequence, the SAME on every rank (the SPMD contract): synchronous blocks, requested sums
    # left queued, blocks WITHOUT a request in between (a gap closes the group), fetched and un-fetched, groups of
    # 1..16, the pipelined mode on and off, the square variant, syncs, and long blocks that re-stride the bus ring
    # (after which a group of short blocks is "mostly gap" and goes as one grouped launch of per-block sums)
    rng = np.random.default_rng(int(os.environ.get("SMX_FUZZ_SEED", "0x5EED7"), 0))
    pipelined, prev = False, None
    for step in range(int(os.environ.get("SMX_TEST_RANDOM_STEPS", "70"))):
        r = rng.random()
        nf = int(rng.choice([1, 1, 7, 16, 64, 64, 64, 65, 300, 2000]))
        tag = ("random", step, nf)
        if pipelined:
            if r < 0.8:
                bus, _ = bank.run(nf)
                want = np.zeros(nf, np.int32)
                if prev is not None:
                    m = min(nf, len(prev))
                    want[:m] = prev[:m]
                assert np.array_equal(bus, want), tag + ("pipelined",)
                prev = expect(nf)[0]
                res["checks"] += 1
            else:
                bank.set_block_mode(0); bank.sync(); pipelined = False
        elif r < 0.30:
            bus, _ = bank.run(nf)
            assert np.array_equal(bus, expect(nf)[0]), tag + ("sync",)
            res["checks"] += 1
        elif r < 0.65:
            bank.run_async(nf)
            bank.allreduce_async(nf)
            want = expect(nf)[0]
            if rng.random() < 0.4:
                assert np.array_equal(bank.fetch(nf)[0], want), tag + ("async+fetch",)
                res["checks"] += 1
        elif r < 0.73:
            bank.run_async(nf)                          # no sum requested: its bus stays this rank's own
            expect(nf)
        elif r < 0.80:
            bank.set_comm_group(int(rng.integers(1, 17)))
        elif r < 0.85:
            bank.sync()
        elif r < 0.90:
            k = int(rng.integers(1, 4))
            got = bank.run_square(k)
            want = np.array([orc.orc_sum_tick_square(all_inc, all_st, len(all_inc)) for _ in range(k)], np.float32)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), tag + ("square",)
            res["checks"] += 1
        else:
            bank.sync(); bank.set_block_mode(1); pipelined, prev = True, None
    if pipelined:
        bank.set_block_mode(0)
    bank.sync()
    bank.set_comm_group(8)
    # phases of the shard after everything
    _, st = bank.read()
    assert np.array_equal(st, all_st[rank * per:(rank + 1) * per])
    rdzv.barrier()
    bank.close()
    rdzv.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
