"""CPU, world_size 2 over gloo: the multi-GPU decomposition (contiguous voice shards,
per-rank int32 bus, integer sum) reproduces the single-bank result bit for bit.
The per-rank compute here is the oracle (no GPU in this container); on the GPU box the
same decomposition is checked on the device by test_full_size_properties_8m_voices and
the in-library RCCL path by test_rccl_allreduce_path_single_rank."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, frames, blocks, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import oracle
    from synth_tools_amd import sharding, synthetic
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    orc = oracle.load()
    tab = np.array([orc.orc_note_to_inc(n) for n in range(128)], np.uint32)
    inc, state = synthetic.saw_bank(n_total, 0x5EED0005, tab, active_fraction=0.9)
    lo, hi = sharding.shard_range(n_total, rank, world)
    si = np.ascontiguousarray(inc[lo:hi])
    ss = np.ascontiguousarray(state[lo:hi]).copy()
    got = []
    for _ in range(blocks):
        bus, _ = oracle.synth_run(orc, si, ss, frames)
        sharding.allreduce_bus(bus)
        got.append(bus.copy())
    np.save(os.path.join(out_dir, "bus_%d.npy" % rank), np.concatenate(got))
    np.save(os.path.join(out_dir, "state_%d.npy" % rank), ss)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_partition():
    from synth_tools_amd import sharding
    for n, w in ((8 << 20, 8), (1000, 3), (7, 8), (65536, 2)):
        edges = [sharding.shard_range(n, r, w) for r in range(w)]
        assert edges[0][0] == 0 and edges[-1][1] == n
        assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
        for v in (0, n // 3, n - 1):
            r = sharding.voice_owner(v, n, w)
            assert edges[r][0] <= v < edges[r][1]


@pytest.mark.timeout(300)
def test_two_rank_bus_sum_is_bit_exact(tmp_path, orc):
    import torch.multiprocessing as mp
    import oracle
    from synth_tools_amd import synthetic
    n_total, frames, blocks, world = 100001, 64, 3, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, frames, blocks, str(tmp_path)), nprocs=world, join=True)
    tab = np.array([orc.orc_note_to_inc(n) for n in range(128)], np.uint32)
    inc, state = synthetic.saw_bank(n_total, 0x5EED0005, tab, active_fraction=0.9)
    want = []
    for _ in range(blocks):
        bus, _ = oracle.synth_run(orc, inc, state, frames)
        want.append(bus)
    want = np.concatenate(want)
    b0 = np.load(tmp_path / "bus_0.npy")
    b1 = np.load(tmp_path / "bus_1.npy")
    assert np.array_equal(b0, want) and np.array_equal(b1, want)
    st = np.concatenate([np.load(tmp_path / "state_0.npy"), np.load(tmp_path / "state_1.npy")])
    assert np.array_equal(st, state)
