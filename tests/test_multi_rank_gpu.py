"""GPU: the PRODUCT's multi-rank path (in-library RCCL communicator, grouped contiguous all-reduce,
pipelined reduce), one process per rank.  On the one-GPU box the ranks share the device when RCCL
accepts that; when it refuses (duplicate-GPU check) the 2-rank case skips and the 1-rank communicator
still runs every code path (its all-reduce is a local copy)."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _fake_rccl():
    """tests/c/fake_rccl.cpp (a shared-memory test double of the seven RCCL entry points the product calls),
    built on the box with hipcc.  None if it cannot be built."""
    src = os.path.join(HERE, "c", "fake_rccl.cpp")
    out = os.path.join(tempfile.gettempdir(), "smx_fake_rccl_%d.so" % os.getuid())
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        if not os.path.exists(hipcc):
            return None
        p = subprocess.run([hipcc, "-O2", "-shared", "-fPIC", "-o", out, src, "-lrt"], capture_output=True, text=True)
        if p.returncode != 0:
            return None
    return out


def _run(world, devices, voices=5000, timeout=240, preload=None, extra_env=None):
    rdzv = tempfile.mkdtemp(prefix="smx_rdzv_test_")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), SMX_RDZV_DIR=rdzv,
                   SMX_TEST_DEVICE=str(devices[r]), SMX_TEST_VOICES=str(voices), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if preload:
            env["LD_PRELOAD"] = preload
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "mgpu_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    out = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=timeout)
            out.append((p.returncode, o, e))
    except subprocess.TimeoutExpired:
        for p in procs:
            p.kill()
        raise
    return out


def test_one_rank_communicator_all_modes(smx):
    (rc, o, e), = _run(1, [0])
    assert rc == 0, e[-2000:]
    res = json.loads(o.strip().splitlines()[-1])
    assert res["ranks_seen"] == 1 and res["checks"] >= 20


@pytest.mark.parametrize("voices", [5000, 1 << 20])
def test_two_ranks_sum_their_shards(smx, voices):
    ndev = smx.lib().smx_device_count()
    devices = [0, 1] if ndev >= 2 else [0, 0]
    res = _run(2, devices, voices=voices)
    errs = "\n".join(e[-1500:] for _, _, e in res)
    if any(rc != 0 for rc, _, _ in res) and ndev < 2 and \
            any(s in errs.lower() for s in ("duplicate gpu", "invalid usage", "invalid argument", "unhandled")):
        pytest.skip("RCCL refuses two ranks on one device here; the >= 2-rank product path needs >= 2 GPUs")
    for rc, o, e in res:
        assert rc == 0, e[-2000:]
        r = json.loads(o.strip().splitlines()[-1])
        assert r["ranks_seen"] == 2 and r["checks"] >= 20


@pytest.mark.parametrize("world,voices", [(2, 5000), (4, 70000), (2, 1 << 20)])
def test_ranks_sharing_the_gpu_through_the_rccl_test_double(smx, world, voices):
    """2 and 4 PROCESSES, each with its shard of the bank on the one GPU, the library's communicator path end to
    end, with tests/c/fake_rccl.cpp standing in for librccl (RCCL itself refuses two ranks on one device).  The
    double sums through shared memory AND checks that every rank issues the same collectives with the same counts.
    Every mode that hands out a reduced bus must equal the oracle run over ALL shards."""
    fake = _fake_rccl()
    if fake is None:
        pytest.skip("hipcc not available to build the RCCL test double")
    res = _run(world, [0] * world, voices=voices, preload=fake)
    for rc, o, e in res:
        assert rc == 0, e[-3000:]
        r = json.loads(o.strip().splitlines()[-1])
        assert r["ranks_seen"] == world and r["checks"] >= 20
        assert r["grouped_block_sums"] == 20 and r["grouped_collectives"] <= 5


def test_bench_with_two_ranks_on_the_one_gpu(smx):
    """`python bench.py --gpus 2` end to end on the one-GPU box: its own launcher, the socket rendezvous, the
    library's communicator (the RCCL test double underneath), the timed loop with one collective per 8 steps, and
    the bus check SUMMED OVER BOTH RANKS against the closed form.  The rate it prints means nothing here (the double
    blocks); that the line appears with ranks_seen = 2 and the check passed is the point."""
    fake = _fake_rccl()
    if fake is None:
        pytest.skip("hipcc not available to build the RCCL test double")
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(LD_PRELOAD=fake, SMX_BENCH_DEVICES="0,0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--voices", str(1 << 20),
                        "--frames", "64", "--steps", "40", "--warmup", "5"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2
    assert "summed over 2 ranks" in line["verified"]
    assert line["config"]["voices_total"] == 2 << 20
    # 45 blocks + the checked one; one collective per 8 blocks (+ the ring's wrap and the final flushes)
    assert line["collectives"]["block_sums_carried"] == 46 and line["collectives"]["issued"] <= 10


def test_the_test_double_catches_a_broken_spmd_contract(smx):
    """include/synth_mi355x.h: with a communicator every rank makes the same sequence of calls.  A rank that
    fetches in the middle of a group on its own would issue other collectives than its peers -- real RCCL would
    hang or mis-sum; the test double names it."""
    fake = _fake_rccl()
    if fake is None:
        pytest.skip("hipcc not available to build the RCCL test double")
    res = _run(2, [0, 0], preload=fake, extra_env={"SMX_TEST_VIOLATE": "1"}, timeout=200)
    assert any(rc != 0 for rc, _, _ in res)
    assert any("SPMD contract" in e or "different collectives" in e for _, _, e in res)
    assert not any("violation_went_unnoticed" in o for _, o, _ in res if o)


@pytest.mark.parametrize("nranks", [2, 3])
def test_plain_c_host_shards_a_bank_over_forked_processes(smx, nranks):
    """tests/c/test_multi_rank.c: fork before any GPU call, the unique id over pipes, smx_bank_comm_init, sync and
    pipelined smx_bank_run on every rank; all ranks hold the same samples, and they are the samples of the whole
    bank in one process.  The header's multi-GPU section used from C, no Python in the data path."""
    fake = _fake_rccl()
    exe = os.path.join(os.path.dirname(HERE), "host", "test_multi_rank.dynamic.host.elf")
    if fake is None or not os.path.exists(exe):
        pytest.skip("needs hipcc (RCCL test double) and the host programs (make -C host)")
    p = subprocess.run([exe, str(nranks), "1"], env=dict(os.environ, LD_PRELOAD=fake), capture_output=True, text=True,
                       timeout=200)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "%d ranks ok" % nranks in p.stderr


@pytest.mark.parametrize("world,voices", [(1, 128), (2, 128), (3, 64), (2, 4096)])
def test_global_allocator_over_sharded_banks(smx, world, voices):
    """"note-on routing is host-side".  smx_bank_shard: every rank runs the reference's allocator over
    the whole global bank (first free voice in GLOBAL voice order, steal voice 0 when the global bank is full, a stray
    note-off silences global voice 0) and applies only its own voices; a collective load exchanges the free maps.
    Against the oracle's allocator over ONE array of world x voices voices: every block's reduced bus, and each
    rank's slice of the final inc[] / state[]."""
    fake = _fake_rccl()
    if fake is None:
        pytest.skip("hipcc not available to build the RCCL test double")
    res = _run(world, [0] * world, voices=voices, preload=fake, extra_env={"SMX_TEST_ALLOCATOR": "1"})
    for rc, o, e in res:
        assert rc == 0, e[-3000:]
        r = json.loads(o.strip().splitlines()[-1])
        assert r["ranks_seen"] == world and r["checks"] >= 20
        if voices <= 128:
            assert r["bank_filled_up"]                       # so the steal-voice-0 path ran
