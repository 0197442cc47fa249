"""GPU: BASELINE config 5's per-GPU shard (1 Mi voices) driven through the PRODUCT's communicator path in
SYNC mode (rank 0 of 1): every block is run, reduced (flush on fetch: one collective per block) and fetched, as a
JACK callback on an 8-GPU node would do it.  Checks the bits against the oracle and records what the
flush-on-fetch costs per block next to the plain (no communicator) sync call: gpurun_out/c5_sync_cost.json."""
import json
import os
import time

import numpy as np
import pytest

import oracle
from synth_tools_amd import synthetic

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _time_blocks(bank, nf, reps):
    for _ in range(20):
        bank.run(nf)
    t0 = time.perf_counter()
    for _ in range(reps):
        bank.run(nf)
    return (time.perf_counter() - t0) / reps * 1e6


def test_c5_shard_sync_mode_through_the_communicator(smx, orc, inc_table):
    n = 1 << 20
    inc, state = synthetic.saw_bank(n, 0x5EED0005, inc_table)
    res = {}
    for with_comm in (False, True):
        bank = smx.SawBank(n)
        bank.load(inc, state)
        if with_comm:
            bank.comm_init(0, 1, smx.comm_unique_id())
            assert bank.comm_ranks() == 1
        st = state.copy()
        for nf in (64, 1, 64):
            bus, vec = bank.run(nf)                     # with a communicator: kernel + all-reduce + fetch
            obus, ovec = oracle.synth_run(orc, inc, st, nf)
            assert np.array_equal(bus, obus) and np.array_equal(vec.view(np.uint32), ovec.view(np.uint32))
        for nf in (1, 64):
            res["%s_%dframes_us_per_block" % ("comm1" if with_comm else "plain", nf)] = round(_time_blocks(bank, nf, 300), 2)
        if with_comm:
            coll, sums = bank.comm_stats()
            assert coll == sums                          # sync mode: one collective per block (flush on fetch)
            res["collectives"] = coll
            # pipelined: the reduce and the copy run behind the kernel on the second stream
            bank.sync()
            bank.set_block_mode(1)
            for nf in (1, 64):
                res["comm1_pipelined_%dframes_us_per_block" % nf] = round(_time_blocks(bank, nf, 300), 2)
            bank.set_block_mode(0)
        bank.close()
    res["note"] = ("wall time of smx_bank_run per block, 1 Mi voices (config 5's shard), one MI355X; comm1 = a 1-rank RCCL "
                   "communicator (its all-reduce is a local copy: this is the cost of the code path, NOT of 8-rank xGMI latency)")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "c5_sync_cost.json"), "w"), indent=1)
    # the real-time budget of a 64-frame block at 48 kHz is 1333 us: the code path must be far below it
    assert res["comm1_64frames_us_per_block"] < 400
