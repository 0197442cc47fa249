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
    # phases of the shard after everything
    _, st = bank.read()
    assert np.array_equal(st, all_st[rank * per:(rank + 1) * per])
    rdzv.barrier()
    bank.close()
    rdzv.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
