#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X saw voice bank (the hot path of
linux/synth.c:169-202 widened to N voices), one JSON line on rank 0.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over the whole resident bank: one
synth_run() block of --frames frames for --voices voices per GPU, state in HBM
before the timed region starts and left there after it.  Voices shard
contiguously over ranks (weak scaling: --voices is per GPU); each rank mixes
its shard to an int32 bus and the buses are summed with one RCCL all-reduce
per step on a second stream (integer sum => bit-exact in any order).

Metric: Gsamples/s = voice-samples advanced per second, whole job.
Roofline: algorithmic HBM bytes per step (8 B read per voice: inc and the lazily
materialised phase base; nothing written back; + 4 B per bus frame) / average kernel time measured with HIP
events on the bank's own stream, against 8 TB/s.
cpu_baseline: the CPU oracle (a port of the reference loop), timed on this
box's host cores on a bounded sample of the same bank (baseline only).
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
INT_VALU_PEAK_TOPS = 256 * 4 * 16 * 2.4e9 / 1e12   # 16 int32 lanes/clk/SIMD (measured: profiles/)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--voices", type=int, default=1 << 26, help="voices per GPU")
    ap.add_argument("--frames", type=int, default=1, help="frames per step (1 = tick(), 64 = JACK process())")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary workloads")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def saw_roofline(voices, frames, kernel_ms):
    # 8 B read per voice (inc + state0; the advanced phase is never written back: DESIGN.md §2)
    alg_bytes = 8.0 * voices + 4.0 * frames
    gbs = alg_bytes / (kernel_ms * 1e-3) / 1e9
    return alg_bytes, gbs


def cpu_baseline_saw(inc, state, frames, budget_s):
    """The oracle's synth_run on the first 2^20 voices of the same bank, 1 core,
    then the same loop with voices split over all cores (not reference behaviour)."""
    import oracle
    orc = oracle.load()
    n = min(len(inc), 1 << 20)
    ci = np.ascontiguousarray(inc[:n])
    cs = np.ascontiguousarray(state[:n]).copy()
    blk = max(frames, 64)
    bus = np.zeros(blk, np.int32)
    t0 = time.perf_counter()
    done = 0
    while True:
        orc.orc_synth_run(ci, cs, n, None, bus.ctypes.data, blk)
        done += blk
        if time.perf_counter() - t0 > budget_s * 0.5:
            break
    dt = time.perf_counter() - t0
    single = n * done / dt / 1e9
    cores = min(len(os.sched_getaffinity(0)), 16)      # the GPU box's CPU share for one GPU
    per = n // cores
    res = [0] * cores

    def work(k):
        ii = np.ascontiguousarray(ci[k * per:(k + 1) * per])
        ss = np.ascontiguousarray(cs[k * per:(k + 1) * per])
        bb = np.zeros(blk, np.int32)
        t = time.perf_counter()
        f = 0
        while time.perf_counter() - t < budget_s * 0.4:
            orc.orc_synth_run(ii, ss, per, None, bb.ctypes.data, blk)
            f += blk
        res[k] = per * f / (time.perf_counter() - t)

    th = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    sample = "first %d voices of the bank, %d-frame blocks, %d frames, gcc -O2 -fwrapv" % (n, blk, done)
    return ({"value": round(single, 4), "unit": "Gsamples/s", "cores": 1, "kind": "port", "sample": sample},
            {"value": round(sum(res) / 1e9, 4), "unit": "Gsamples/s", "cores": cores, "kind": "port",
             "sample": "same bank split over %d threads, partial buses (baseline-parallel, not reference behaviour)" % cores})


def cpu_baselines_extra(synthetic, tab, budget_s=2.0):
    """BASELINE.md §3 rows B1 and B4 on one core: the reference's own operating point (64 voices,
    64-frame blocks) and the carry-out PDM loop (dither 0) on 65 536 channels."""
    import oracle
    orc = oracle.load()
    out = []
    inc = tab[30:94].astype(np.uint32).copy()                    # notes 30..93 all on
    st = np.zeros(64, np.uint32)
    bus = np.zeros(64, np.int32)
    t0, blocks = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s:
        for _ in range(2000):
            orc.orc_synth_run(inc, st, 64, None, bus.ctypes.data, 64)
        blocks += 2000
    dt = time.perf_counter() - t0
    out.append({"workload": "B1: reference operating point, 64 voices x 64-frame blocks (linux/synth.c), 1 core",
                "value": round(64 * 64 * blocks / dt / 1e9, 4), "unit": "Gsamples/s", "cores": 1, "kind": "port"})
    sp, ac = synthetic.pdm_bank(65536, 0x5EED0003)
    words = 65536 // 32
    bits = np.zeros(64 * words, np.uint32)
    t0, ticks = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s:
        orc.orc_pdm_run(sp, ac, 65536, None, 64, bits)
        ticks += 64
    dt = time.perf_counter() - t0
    out.append({"workload": "B4: carry-out PDM bank (mod_pdm.c), 65536 channels, dither 0, 1 core",
                "value": round(65536 * ticks / dt / 1e9, 4), "unit": "Gsamples/s (channel-ticks)", "cores": 1,
                "kind": "port"})
    return out


def time_saw(sta, bank, frames, steps, warmup, comm=False, settle_ms=10.0):
    """Average ms per step over `steps` steps.  The secondary workloads follow read-backs and CPU
    work that leave the GPU idle, and the first milliseconds after an idle gap run up to 15 % slow
    (clock ramp): besides the `warmup` steps, untimed steps are run until `settle_ms` have passed."""
    t0 = time.perf_counter()
    for _ in range(warmup):
        bank.run_async(frames)
        if comm:
            bank.allreduce_async(frames)
    bank.sync()
    while (time.perf_counter() - t0) * 1e3 < settle_ms:
        for _ in range(max(1, warmup)):
            bank.run_async(frames)
        bank.sync()
    bank.timer_start()
    for _ in range(steps):
        bank.run_async(frames)
        if comm:
            bank.allreduce_async(frames)
    ms = bank.timer_stop()
    bank.sync()
    return ms / steps


def settle(step, sync, ms=10.0):
    """Untimed steps until `ms` have passed (see time_saw): clock ramp after an idle gap."""
    t0 = time.perf_counter()
    while True:
        step()
        sync()
        if (time.perf_counter() - t0) * 1e3 >= ms:
            return


def also_workloads(sta, synthetic, tab, big_bank, voices):
    out = []
    # the JACK operating point (48 kHz, 64 frames: linux/jack_midi.c:19-20) on the same bank
    for frames in (16, 32, 64, 1024):
        ms = time_saw(sta, big_bank, frames, 50 if frames < 1024 else 5, 5 if frames < 1024 else 1)
        _, gbs = saw_roofline(voices, frames, ms)
        vs = voices * frames / (ms * 1e-3)
        out.append({"workload": "saw bank, %d voices, %d frames/step" % (voices, frames),
                    "value": round(vs / 1e9, 2), "unit": "Gsamples/s", "ms_per_step": round(ms, 5),
                    "hbm_GBs": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4),
                    # vector ops per voice-sample: 2.5 (direct) / 1.5 + 1 scalar (carry formulation, stepping;
                    # above 32 frames the device picks between stepping and locating the wraps: DESIGN 3.2b)
                    "formulation": ("carry (stepping / wrap events, picked on the device)" if frames > 32 else "carry (stepping)")
                                   if frames > 16 and voices * frames >= 1 << 30 else "direct",
                    # share of the int32 vector issue rate at the stepping forms' instructions per
                    # voice-sample; not defined when the wraps are located instead of stepped
                    "int_valu_frac": None if frames > 32 and voices * frames >= 1 << 30 else
                                     round(vs * (1.5 if frames > 16 and voices * frames >= 1 << 30 else 2.5)
                                           / 1e12 / INT_VALU_PEAK_TOPS, 4),
                    "max_voices_48k": int(vs / 48000)})
    # the same 64-frame blocks on a bank of high voices only (MIDI notes 100..127: 3..17 wraps per voice
    # per block): the device-side statistic keeps the stepping form, whose time does not depend on the data
    r = synthetic.splitmix64(0x5EED0009, voices)
    hi_inc = np.ascontiguousarray(tab[100 + (r % np.uint64(28)).astype(np.int64)].astype(np.uint32))
    big_bank.load(inc=hi_inc)
    ms = time_saw(sta, big_bank, 64, 50, 5)
    out.append({"workload": "saw bank, %d voices, 64 frames/step, notes 100..127 only (stepping form picked on the device)" % voices,
                "value": round(voices * 64 / (ms * 1e-3) / 1e9, 2), "unit": "Gsamples/s", "ms_per_step": round(ms, 5)})
    # BASELINE config 2: 65 536 voices, 64-frame blocks
    inc, st = synthetic.saw_bank(65536, 0x5EED0002, tab)
    b = sta.SawBank(65536)
    b.load(inc, st)
    ms = time_saw(sta, b, 64, 200, 20)
    b.close()
    out.append({"workload": "c2: saw bank, 65536 voices, 64 frames/step (launch-bound)",
                "value": round(65536 * 64 / (ms * 1e-3) / 1e9, 2), "unit": "Gsamples/s",
                "ms_per_step": round(ms, 5)})
    b = sta.SawBank(65536)
    b.load(inc, st)
    ms = time_saw(sta, b, 4096, 50, 5)
    b.close()
    out.append({"workload": "c2: saw bank, 65536 voices, 4096 frames/launch (64 JACK blocks per launch, time-parallel chunks)",
                "value": round(65536 * 4096 / (ms * 1e-3) / 1e9, 2), "unit": "Gsamples/s",
                "ms_per_step": round(ms, 5)})
    # BASELINE config 5's per-GPU shard: 8 Mi voices over 8 GPUs = 1 Mi voices each
    inc, st = synthetic.saw_bank(1 << 20, 0x5EED0005, tab)
    b = sta.SawBank(1 << 20)
    b.load(inc, st)
    for frames in (1, 64):
        ms = time_saw(sta, b, frames, 200, 20)
        out.append({"workload": "c5 shard: saw bank, 1048576 voices (1/8 of 8 Mi), %d frame(s)/step" % frames,
                    "value": round((1 << 20) * frames / (ms * 1e-3) / 1e9, 2), "unit": "Gsamples/s",
                    "ms_per_step": round(ms, 5)})
    b.close()
    # BASELINE config 3: 1 Mi PDM channels (mod_pdm.c integer path), dither 0 and seeded
    n, nt = 1 << 20, 4096
    sp, ac = synthetic.pdm_bank(n, 0x5EED0003)
    p = sta.PdmBank(n)
    p.load(sp, ac)
    for with_d in (False, True):
        if with_d:
            # seeded dither for the perf leg only: parity tests cover explicit dither arrays
            p.tick_n(64, synthetic.dither_stream(64, 7, 0x0FFFFFFF), want_bits=False)
        settle(lambda: p.tick_n_async(nt, with_d), p.sync)
        p.timer_start()
        reps = 20
        for _ in range(reps):
            p.tick_n_async(nt, with_d)
        ms = p.timer_stop() / reps
        alg = 12.0 * n + nt * n / 8.0
        out.append({"workload": "c3: carry-out PDM bank, %d channels, %d ticks/launch, dither=%s" % (n, nt, "seeded" if with_d else "0"),
                    "value": round(n * nt / (ms * 1e-3) / 1e9, 2), "unit": "Gsamples/s (channel-ticks)",
                    "ms_per_step": round(ms, 4), "hbm_GBs": round(alg / (ms * 1e-3) / 1e9, 1),
                    "hbm_frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
    # the same bank with channel-stream output (no transpose: 2 instead of 3 vector ops per tick)
    for with_d in (False, True):
        settle(lambda: p.tick_n_streams_async(nt, with_d), p.sync)
        p.timer_start()
        for _ in range(20):
            p.tick_n_streams_async(nt, with_d)
        ms = p.timer_stop() / 20
        alg = 12.0 * n + nt * n / 8.0
        out.append({"workload": "c3: carry-out PDM bank, %d channels, %d ticks/launch, channel-stream layout, dither=%s" % (n, nt, "seeded" if with_d else "0"),
                    "value": round(n * nt / (ms * 1e-3) / 1e9, 2), "unit": "Gsamples/s (channel-ticks)",
                    "ms_per_step": round(ms, 4), "hbm_GBs": round(alg / (ms * 1e-3) / 1e9, 1),
                    "hbm_frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
    p.close()
    # noise-shaped PWM bank (mod_pdm_pwm.c: pdm2 + glide + control rate), 1 Mi channels
    n, nt = 1 << 20, 1024
    w = sta.PwmBank(n, order=2)
    w.load(setpoint=synthetic.pdm_bank(n, 0x5EED0008)[0])
    w.tick_n(8, synthetic.dither_stream(8, 7, 0x3FF), want_duty=False)
    settle(lambda: w.tick_n_async(nt, True), w.sync)
    w.timer_start()
    for _ in range(20):
        w.tick_n_async(nt, True)
    ms = w.timer_stop() / 20
    w.close()
    alg = 52.0 * n + float(nt) * n
    out.append({"workload": "noise-shaped PWM bank (pdm2+glide), %d channels, %d ticks/launch, dither seeded" % (n, nt),
                "value": round(n * nt / (ms * 1e-3) / 1e9, 2), "unit": "Gsamples/s (channel-ticks)",
                "ms_per_step": round(ms, 4), "hbm_GBs": round(alg / (ms * 1e-3) / 1e9, 1),
                "hbm_frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
    # BASELINE config 4: 256 Ki poly voices (saw + 1-pole LPF + ADSR + stereo mix; build-defined)
    n = 1 << 18
    pb = sta.PolyBank(n)
    pb.load(**synthetic.poly_bank(n, 0x5EED0004, tab))
    settle(lambda: pb.run_async(64), pb.sync)
    pb.timer_start()
    reps = 200
    for _ in range(reps):
        pb.run_async(64)
    ms = pb.timer_stop() / reps
    pb.close()
    alg = 60.0 * n + 64 * 8
    out.append({"workload": "c4: poly bank (saw+LPF+ADSR, stereo), %d voices, 64 frames/step" % n,
                "value": round(n * 64 / (ms * 1e-3) / 1e9, 2), "unit": "Gsamples/s", "ms_per_step": round(ms, 5),
                "hbm_GBs": round(alg / (ms * 1e-3) / 1e9, 1), "hbm_frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
    return out


def main():
    # stdout carries exactly ONE line (the JSON): libraries that print banners at init (RCCL
    # prints its version block to stdout) are pointed at stderr for the whole run.
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py: --gpus %d needs `python -m torch.distributed.run --nproc-per-node %d`" % (a.gpus, a.gpus))
        a.gpus = world

    import torch
    import synth_tools_amd as sta
    from synth_tools_amd import synthetic

    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible (there is no CPU fallback)")
    torch.cuda.set_device(local)
    dist = None
    # SMX_BENCH_FORCE_DIST: rehearse the N > 1 control path (torch process group + in-library RCCL
    # communicator in one process) with a single rank under torch.distributed.run
    force_dist = world == 1 and bool(os.environ.get("SMX_BENCH_FORCE_DIST")) and "MASTER_ADDR" in os.environ
    if world > 1 or force_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    tab = synthetic.note_inc_table(sta.lib().note_to_inc)
    # rank r owns voices [r*V, (r+1)*V) of the global bank: its own splitmix64 stream
    inc, state = synthetic.saw_bank(a.voices, 0x5EED0005 + 0x1000 * rank, tab)
    bank = sta.SawBank(a.voices, device=local)
    bank.load(inc, state)

    if dist:
        uid = torch.zeros(sta.UNIQUE_ID_BYTES, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid.copy_(torch.from_numpy(sta.comm_unique_id()))
        dist.broadcast(uid, 0)
        bank.comm_init(rank, world, uid.cpu().numpy())

    comm = dist is not None
    if not comm and os.environ.get("SMX_BENCH_FORCE_COMM"):
        # rehearsal of the multi-GPU code path on one GPU: 1-rank RCCL communicator
        bank.comm_init(0, 1, sta.comm_unique_id())
        comm = True
    for _ in range(a.warmup):
        bank.run_async(a.frames)
        if comm:
            bank.allreduce_async(a.frames)
    bank.sync()

    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bank.timer_start()
    for _ in range(a.steps):
        bank.run_async(a.frames)
        if comm:
            bank.allreduce_async(a.frames)
    kernel_ms = bank.timer_stop() / a.steps        # HIP events on the kernel's stream
    bank.sync()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt, kernel_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, kernel_ms = float(t[0]), float(t[1])

    # sanity on the timed state: phases advanced by exactly (warmup+steps)*frames*inc
    _, gst = bank.read()
    total_frames = (a.warmup + a.steps) * a.frames
    ok = bool(np.array_equal(gst[:1 << 16], (state[:1 << 16] + np.uint32(total_frames) * inc[:1 << 16])))
    if not ok:
        sys.exit("bench.py: phase check failed after the timed region")

    if rank == 0:
        vs = world * a.voices * a.frames * a.steps / dt
        alg_bytes, gbs = saw_roofline(a.voices, a.frames, kernel_ms)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            key = "saw_v%d_f%d" % (a.voices, a.frames)
            if key in tj:
                traffic = tj[key]["hbm_bytes_per_launch"]
        line = {
            "metric": "Gsamples/s (voice-samples/s, saw oscillator bank)",
            "value": round(vs / 1e9, 3), "unit": "Gsamples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "saw voice bank resident in HBM, %d voices/GPU x %d frame(s)/step "
                                   "(tick ABI), seeded note bank (splitmix64), all voices on" % (a.voices, a.frames),
                       "voices_per_gpu": a.voices, "frames_per_step": a.frames,
                       "voices_total": world * a.voices,
                       "baseline_config": "BASELINE configs[1] (int32 phase-accumulator saw voices on 1 MI355X, "
                                          "bit-exact) scaled from 65 536 voices (512 KiB, launch-bound: see `also`) to an "
                                          "HBM-resident bank, the regime the %HBM-roofline metric is defined in",
                       "parallelism": "voices sharded x%d, int32 bus all-reduce (RCCL)" % world if comm else "1 GPU"},
            "max_voices_48k": int(vs / 48000),
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "saw_bank_kernel", "kernel_ms": round(kernel_ms, 5),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "bytes_per_voice": 8,
                         "note": "8 B read per voice per launch (inc + phase base); the advanced phase is kept as "
                                 "state0 + elapsed*inc and never written back, so SURVEY 8d's 4-byte state write "
                                 "(12 B/voice) does not exist on this path; PMC traffic agrees (profiles/)"},
        }
        if world == 1 and not a.no_cpu:
            single, par = cpu_baseline_saw(inc, state, a.frames, a.cpu_seconds)
            line["cpu_baseline"] = single
            line["cpu_baseline_parallel"] = par
            line["cpu_baselines_extra"] = cpu_baselines_extra(synthetic, tab)
        if world == 1 and not a.no_also:
            line["also"] = also_workloads(sta, synthetic, tab, bank, a.voices)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    if dist:
        dist.barrier()          # every rank is done with its communicator before any is torn down
    bank.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
