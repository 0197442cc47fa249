#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X saw voice bank (the hot path of
linux/synth.c:169-202 widened to N voices), one JSON line on rank 0.

    python bench.py --gpus N --steps K --warmup W

One process per GPU.  `--gpus N` with N > 1 starts its N ranks itself (fresh child processes,
before anything touches a GPU); under `python -m torch.distributed.run --nproc-per-node N ...`
the ranks it is given (RANK / LOCAL_RANK / WORLD_SIZE) are used as they are.  No torch in either
case: the library creates its own RCCL communicator from a 128-byte id that the ranks exchange
over a localhost socket (synth_tools_amd/rendezvous.py), which also carries the barrier and the
max over ranks of the timing.

A "step" is one pass of the hot path over the whole resident bank: one synth_run() block of
--frames frames for --voices voices per GPU, state in HBM before the timed region starts and
left there after it.  Voices shard contiguously over ranks (weak scaling: --voices is per GPU);
each rank mixes its shard to an int32 bus and the buses are summed by RCCL all-reduces on a second
stream (integer sum => bit-exact in any order; one collective per 8 steps).

Metric: Gsamples/s = voice-samples advanced per second, whole job.
Roofline: algorithmic HBM bytes per step (8 B read per voice: inc and the lazily materialised
phase base; nothing written back; + 4 B per bus frame) / average kernel time
measured with HIP events on the bank's own stream, against 8 TB/s.
After the timed region one more step runs and its BUS (the timed kernel's only output) is
compared with the closed form sum_v ((int32)(state0 + T*inc) >> 4) computed in numpy; a mismatch
exits non-zero.  The secondary workloads are checked the same way.
cpu_baseline: the reference's own linux/synth.c (compiled verbatim into oracle/_ref; "reference")
or the oracle's port of it ("port"), timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# Vector issue: measured cycles (at 2.4 GHz) one SIMD spends per wave64 instruction at full occupancy
# (tools/ubench/valu_rates.hip, profiles/r02_valu_rates.txt): v_add/v_sub/v_ashrrev/v_and/v_xor/v_mov ~2.7,
# everything VOP3-only or carry-writing (v_add3, v_add_co, v_addc_co, v_mul_lo, v_perm, v_cndmask ...) ~4.5.
# The kernels' inner loops in those units, per 64 units of work (one wave instruction each lane-op):
SIMD_CYCLES_PER_S = 256 * 4 * 2.4e9
ISSUE = {"saw_direct": 2.66 + 2.85 + 2.85,            # v_ashrrev + v_add (phase) + v_add (accumulate) per voice-sample
         "saw_carry": (4 * 4.50 + 2 * 4.64) / 4,      # 4 v_mad_u64_u32 ({wraps, phase} += inc) + 2 v_add3 per 4 voice-samples
         "pdm_tick_major": 10.0,                      # v_add_co + s_nop 1 + 2 v_writelane, measured as a sequence
         "pdm_stream": 4.46 + 4.41,                   # v_add_co + v_addc_co
         "dither_add": 2.85,
         "pwm2": 2.48 + 2.66 + 2.85 + 2 * 4.72 + 0.75 * 4.28,   # and, sub, add, 2 add3, 3 v_perm per 4 channel-ticks
         "poly": None}                                # mixed int / fp32 / LDS: see the counters in profiles/
ALL_LEGS = ("saw_frames", "saw_hi", "c2", "c5", "c3", "c3_streams", "pwm", "c4", "pdm_tick")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--voices", type=int, default=1 << 26, help="voices per GPU")
    ap.add_argument("--frames", type=int, default=1, help="frames per step (1 = tick(), 64 = JACK process())")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary workloads")
    ap.add_argument("--legs", default=",".join(ALL_LEGS), help="secondary workloads to run: " + ",".join(ALL_LEGS))
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--no-verify", action="store_true", help="skip the bus checks (profiling runs)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--repeats", type=int, default=9,
                    help="the timed region of --steps steps is run this many times; the line reports the median")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks ourselves, before any GPU call
# ---------------------------------------------------------------------------------------------
def launch_ranks(n):
    """Parent of a `python bench.py --gpus N` run: N fresh children (never an exec of a process that
    touched the GPU; this one never does), rank 0 prints the JSON line on our stdout."""
    rdzv = tempfile.mkdtemp(prefix="smx_rdzv_")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT="0", SMX_RDZV_DIR=rdzv)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    live = set(range(n))
    failed_at = None
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                failed_at = time.monotonic()
                sys.stderr.write("bench.py: rank %d exited with %d\n" % (r, code))
        # a failed rank leaves the others waiting at the rendezvous or in a collective: give them a moment to
        # fail by themselves (and say why), then stop exactly the children we started
        if failed_at is not None and live and time.monotonic() - failed_at > 3.0:
            sys.stderr.write("bench.py: stopping the other ranks\n")
            for o in sorted(live):
                procs[o].terminate()
            failed_at = float("inf")
        time.sleep(0.05)
    try:
        for f in os.listdir(rdzv):
            os.unlink(os.path.join(rdzv, f))
        os.rmdir(rdzv)
    except OSError:
        pass
    return rc


# ---------------------------------------------------------------------------------------------
# closed forms (numpy) the outputs of the timed kernels are checked against
# ---------------------------------------------------------------------------------------------
def wrap_i32(x):
    return ((np.asarray(x, np.int64) + (1 << 31)) % (1 << 32) - (1 << 31)).astype(np.int64)


def saw_bus_closed_form(inc, state0, t_first, frames):
    """sum_tick_saw (linux/synth.c:169-179) at the frames `t_first + f`: the phasor is linear, so
    phase(t) = state0 + t*inc mod 2^32; int32 wrapping sum of (int)phase >> 4 over the voices that are on."""
    on = inc != 0
    out = np.zeros(len(frames), np.int64)
    with np.errstate(over="ignore"):
        for k, f in enumerate(frames):
            ph = state0 + np.uint32((t_first + int(f)) & 0xFFFFFFFF) * inc
            out[k] = int(np.where(on, ph.view(np.int32) >> 4, 0).sum(dtype=np.int64))
    return wrap_i32(out)


def pdm_rows_closed_form(sp, accu0, dither, rows):
    """Carry-out PDM (mod_pdm.c:214-244, 259-264): accu(t) = accu0 + sum_{k<=t} (sp + d_k) mod 2^32 and the
    pulse of tick t is the carry of that add: accu(t) < (sp + d_t).  -> packed words of the given rows."""
    n = len(sp)
    out = []
    with np.errstate(over="ignore"):
        cum = np.cumsum(dither.astype(np.uint64)) if dither is not None else None
        for t in rows:
            d_t = np.uint32(dither[t]) if dither is not None else np.uint32(0)
            dsum = np.uint32(int(cum[t]) & 0xFFFFFFFF) if dither is not None else np.uint32(0)
            spd = sp + d_t
            acc = accu0 + np.uint32((t + 1) & 0xFFFFFFFF) * sp + dsum
            bits = (acc < spd).astype(np.uint8)
            out.append(np.packbits(bits.reshape(-1, 32), axis=1, bitorder="little").view(np.uint32).reshape(-1))
    return out


def poly_block_numpy(a, nframes):
    """The build-defined poly voice (no reference counterpart) for one block, vectorised
    over voices: saw -> 1-pole LPF (two roundings, never fused) -> integer ADSR -> int stereo mix."""
    inc, phase = a["inc"].copy(), a["phase"].copy()
    y, co = a["y"].copy(), a["a"]
    level, stage = a["level"].copy(), a["stage"].copy()
    on = inc != 0
    g1 = a["gate"] != 0
    stage = np.where(on & g1 & ((stage == 0) | (stage == 4)), 1, stage)
    stage = np.where(on & ~g1 & (stage != 0), 4, stage).astype(np.uint32)
    pl = (a["pan"] & 0xFFFF).astype(np.int64)
    pr = (a["pan"] >> 16).astype(np.int64)
    bus = np.zeros((nframes, 2), np.int64)
    ar, dr, sl, rr = a["ar"], a["dr"], a["sl"], a["rr"]
    with np.errstate(over="ignore"):
        for i in range(nframes):
            x = phase.view(np.int32).astype(np.float32) * np.float32(2.0 ** -31)
            phase = np.where(on, phase + inc, phase)
            t = x - y
            y = np.where(on, y + co * t, y).astype(np.float32)
            nl = level + ar
            a_wrap = nl < level
            lv_a = np.where(a_wrap, np.uint32(0xFFFFFFFF), nl)
            st_a = np.where(a_wrap, 2, 1)
            d_arr = (level <= sl) | ((level - sl) <= dr)
            lv_d = np.where(d_arr, sl, level - dr)
            st_d = np.where(d_arr, 3, 2)
            r_arr = level <= rr
            lv_r = np.where(r_arr, np.uint32(0), level - rr)
            st_r = np.where(r_arr, 0, 4)
            lv = np.select([stage == 1, stage == 2, stage == 3, stage == 4], [lv_a, lv_d, sl, lv_r], np.uint32(0))
            st = np.select([stage == 1, stage == 2, stage == 3, stage == 4], [st_a, st_d, 3, st_r], 0)
            level = np.where(on, lv, level).astype(np.uint32)
            stage = np.where(on, st, stage).astype(np.uint32)
            g = (level >> 8).astype(np.float32) * np.float32(2.0 ** -24)
            o = (y * g).astype(np.float32)
            q = np.where(on, (o * np.float32(524288.0)).astype(np.int32), 0).astype(np.int64)
            bus[i, 0] = int((q * pl).sum())
            bus[i, 1] = int((q * pr).sum())
    return wrap_i32(bus)


def pwm2_block_numpy(st, dither, nt, div_count, div_log=12, sh=24):
    """The firmware's order-2 noise-shaped PWM channel (mod_pdm_pwm.c:108-143 with pdm2_update, pdm.h:32-40, and
    control_update, mod_controlrate.c:28-40) for `nt` ticks, vectorised over the channels of `st` (a dict of uint32
    arrays: setpoint pos0 vel0 pos1 vel1 s1 s2).  -> (duty uint8[nt, n], state after).  All arithmetic mod 2^32."""
    sp, pos0, vel0 = st["setpoint"].copy(), st["pos0"].copy(), st["vel0"].copy()
    pos1, vel1, s1, s2 = st["pos1"].copy(), st["vel1"].copy(), st["s1"].copy(), st["s2"].copy()
    duty = np.zeros((nt, len(sp)), np.uint8)
    div = 1 << div_log
    with np.errstate(over="ignore"):
        for t in range(nt):
            trigger = div_count == 0
            if trigger:                                    # PDM_COPY_LINE, mod_pdm_pwm.c:118-119
                pos0, vel0 = pos1.copy(), vel1.copy()
            pos0 = pos0 + vel0                             # glide :95-98
            q = s2 >> np.uint32(sh)                        # pdm2_update: the output is the quantised LAST state
            a = (q << np.uint32(sh)) + np.uint32(dither[t])
            s1 = s1 + (pos0 - a)
            s2 = s2 + (s1 - a)
            duty[t] = q.astype(np.uint8)
            div_count = (div_count + 1) % div
            if trigger:                                    # control_update runs after the tick that triggered it
                pos1 = pos1 + (vel1 << np.uint32(div_log))
                vel1 = ((sp - pos1).view(np.int32) >> div_log).view(np.uint32)
    return duty, {"setpoint": sp, "pos0": pos0, "vel0": vel0, "pos1": pos1, "vel1": vel1, "s1": s1, "s2": s2}


def roof(alg_bytes, ms, units=None, issue_cycles=None):
    """Roofline block of one launch: algorithmic HBM bytes against 8 TB/s, and the issue time of the inner
    loop's vector instructions (`units` units of work at `issue_cycles` SIMD cycles per 64 of them, all 1024
    SIMDs busy) as a fraction of the launch; `bound` names the nearer ceiling."""
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    r = {"algorithmic_bytes": float(alg_bytes), "kernel_ms": round(ms, 5), "hbm_GBs": round(gbs, 1),
         "hbm_frac": round(gbs / HBM_PEAK_GBS, 4), "valu_issue_frac": None, "bound": "hbm"}
    if units and issue_cycles:
        r["valu_issue_frac"] = round(units / 64.0 * issue_cycles / SIMD_CYCLES_PER_S / (ms * 1e-3), 4)
        if r["valu_issue_frac"] > r["hbm_frac"]:
            r["bound"] = "vector issue"
    return r


_COUNTERS = None


def recorded_issue(kernel_has, ms, cycles_per_inst):
    """`valu_issue_frac` for a kernel whose loop has no closed instruction count (data-dependent trip counts, or a mix
    of integer / fp32 / LDS work): the vector instructions per launch RECORDED by tools/prof_counters.sh
    (profiles/counters.json: SQ_INSTS_VALU of one rocprofv3 --pmc pass over this bench) x the mean issue cost of the
    loop's instruction mix (profiles/r02_valu_rates.txt), against the launch time measured now.  -> (fraction, note)
    or (None, None) when the kernel is not in the record."""
    global _COUNTERS
    if _COUNTERS is None:
        path = os.path.join(ROOT, "profiles", "counters.json")
        _COUNTERS = json.load(open(path)) if os.path.exists(path) else {}
    for name, c in _COUNTERS.get("kernels", {}).items():
        if all(k in name for k in kernel_has) and c.get("SQ_INSTS_VALU"):
            frac = c["SQ_INSTS_VALU"] / 1024.0 * cycles_per_inst / (2.4e9 * ms * 1e-3)
            return round(frac, 4), ("recorded: %s -- %.0f vector instructions per launch x %.2f SIMD cycles each (mean of the "
                                    "loop's mix, profiles/r02_valu_rates.txt), not measured in this run"
                                    % (_COUNTERS.get("source", "profiles/counters.json"), c["SQ_INSTS_VALU"], cycles_per_inst))
    return None, None


def with_recorded_issue(e, kernel_has, cycles_per_inst):
    frac, note = recorded_issue(kernel_has, e["ms_per_step"], cycles_per_inst)
    if frac is not None:
        e["roofline"]["valu_issue_frac"] = frac
        e["roofline"]["valu_issue_source"] = note
        e["roofline"]["bound"] = "vector issue" if frac > e["roofline"]["hbm_frac"] else "hbm"
    return e


def saw_roofline(voices, frames, kernel_ms):
    # 8 B read per voice (inc + state0; the advanced phase is never written back: DESIGN.md 2)
    alg_bytes = 8.0 * voices + 4.0 * frames
    gbs = alg_bytes / (kernel_ms * 1e-3) / 1e9
    return alg_bytes, gbs


# ---------------------------------------------------------------------------------------------
# CPU baselines (the only place the oracle package is used)
# ---------------------------------------------------------------------------------------------
def cpu_baseline_reference(inc, state, budget_s):
    """The reference itself: linux/synth.c:27-208 compiled verbatim (oracle/_ref/libref_synth.so), its
    64-voice struct synth filled with the first 64 voices of the bank, synth_run in 64-frame blocks."""
    import ctypes as C
    import oracle
    ref = oracle.load_ref_synth()
    if ref is None:
        return None
    x = oracle.RefSynth()
    ref.synth_init(C.byref(x))
    for v in range(64):
        x.voice[v].note_inc = int(inc[v])
        x.voice[v].note_state = int(state[v])
    vec = np.zeros(64, np.float32)
    t0, blocks = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s:
        for _ in range(2000):
            ref.synth_run(C.byref(x), vec, 64)
        blocks += 2000
    dt = time.perf_counter() - t0
    return {"value": round(64 * 64 * blocks / dt / 1e9, 4), "unit": "Gsamples/s", "cores": 1, "kind": "reference",
            "sample": "linux/synth.c:27-208 compiled verbatim (gcc -O2 -fwrapv), its 64 voices = the first 64 voices "
                      "of the bank, synth_run in 64-frame blocks, %d blocks" % blocks}


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
    """BASELINE.md 3 row B4 on one core: the carry-out PDM loop (dither 0) on 65 536 channels."""
    import oracle
    orc = oracle.load()
    out = []
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


# ---------------------------------------------------------------------------------------------
# timing helpers
# ---------------------------------------------------------------------------------------------
class Saw:
    """A saw bank plus what the bus check needs: the loaded arrays and the frames run since."""

    def __init__(self, sta, voices, inc, state, device=0):
        self.bank = sta.SawBank(voices, device=device)
        self.bank.load(inc, state)
        self.inc, self.state0, self.t = inc, state, 0
        self.voices = voices

    def run_async(self, frames):
        self.bank.run_async(frames)
        self.t += frames

    def rebase(self):
        """After a reload of the increments: phases as the bank holds them now."""
        self.inc, self.state0 = self.bank.read()
        self.t = 0

    def verify(self, frames, what, rdzv=None, comm=False, pick=None):
        """One more block; its bus (summed over ranks when comm) against the closed form."""
        t_first = self.t
        self.run_async(frames)
        if comm:
            self.bank.allreduce_async(frames)
        bus, _ = self.bank.fetch(frames)
        pick = list(range(frames)) if pick is None else pick
        want = saw_bus_closed_form(self.inc, self.state0, t_first, pick)
        if rdzv is not None and rdzv.world > 1:
            parts = rdzv.allgather(want.astype("<i8").tobytes())
            want = wrap_i32(sum(np.frombuffer(p, "<i8") for p in parts))
        ok = bool(np.array_equal(bus[pick].astype(np.int64), want))
        if rdzv is not None:
            ok = rdzv.all_ok(ok)
        if not ok:
            sys.exit("bench.py: BUS CHECK FAILED (%s): the timed kernel's output differs from the closed form" % what)
        return len(pick)


def time_saw(saw, frames, steps, warmup, comm=False, settle_ms=30.0):
    """Average ms per step over `steps` steps.  The secondary workloads follow read-backs and CPU
    work that leave the GPU idle, and the first 15-25 ms after an idle gap or a change of kernel run
    up to 30 % slow (tools/explore_settle.py: 64 Mi voices x 16 frames 130 -> 101 us over the first
    120 launches): besides the `warmup` steps, untimed steps are run until `settle_ms` have passed."""
    bank = saw.bank
    t0 = time.perf_counter()
    for _ in range(warmup):
        saw.run_async(frames)
        if comm:
            bank.allreduce_async(frames)
    bank.sync()
    settle(lambda: saw.run_async(frames), bank.sync, ms=max(0.0, settle_ms - (time.perf_counter() - t0) * 1e3))
    bank.timer_start()
    for _ in range(steps):
        saw.run_async(frames)
        if comm:
            bank.allreduce_async(frames)
    ms = bank.timer_stop()
    bank.sync()
    return ms / steps


def settle(step, sync, ms=30.0, cap_ms=600.0):
    """Untimed steps until the launch time has settled: at least `ms` (clock ramp after an idle gap: time_saw) AND
    until two consecutive batches of steps take the same time within 4 %.  The second condition is there because a
    large device-to-host read-back (a leg's check of a 64 Mi-element bank) leaves the NEXT launches several times
    slower for tens of milliseconds -- longer than any fixed pause we tried (measured: the first 20 launches of a
    1 Mi-channel PWM block 0.83 ms each after such a read, 0.21 ms from then on; tools/README.md).  Bounded by cap_ms."""
    t0 = time.perf_counter()
    last = None
    while True:
        b0 = time.perf_counter()
        for _ in range(8):
            step()
        sync()
        now = time.perf_counter()
        batch = now - b0
        spent = (now - t0) * 1e3
        if spent >= cap_ms:
            return
        if spent >= ms and last is not None and abs(batch - last) <= 0.04 * max(batch, last):
            return
        last = batch


def saw_entry(workload, voices, frames, ms, issue, extra=None):
    vs = voices * frames / (ms * 1e-3)
    e = {"workload": workload, "value": round(vs / 1e9, 2), "unit": "Gsamples/s", "ms_per_step": round(ms, 5),
         "max_voices_48k": int(vs / 48000),
         "roofline": roof(8.0 * voices + 4.0 * frames, ms, voices * frames, ISSUE[issue] if issue else None),
         "verified": "bus == closed form"}
    e["hbm_frac"] = e["roofline"]["hbm_frac"]
    if extra:
        e.update(extra)
    return e


# ---------------------------------------------------------------------------------------------
# secondary workloads: every BASELINE config on one GPU, each with its own roofline block
# ---------------------------------------------------------------------------------------------
def also_workloads(sta, synthetic, tab, big, voices, legs, verify):
    out = []
    SMX_FORM_AUTO, SMX_FORM_STEPPING = 0, 1
    if "saw_frames" in legs:
        # longer blocks of the same bank; 64 frames = the JACK operating point (linux/jack_midi.c:19-20)
        for frames in (8, 16, 32, 64, 128, 1024):
            # (blocks of 17..32 frames of >= 2^25-voice banks run as one 32-frame chunk of the same two forms;
            # with STEPPING pinned they keep the direct form: saw_bank.hip, launch_saw_bank)
            short = 16 < frames <= 32 and voices >= 1 << 25
            carry = (frames > 32 and voices * frames >= 1 << 30) or short
            # such blocks have two exact forms: AUTO (default) lets the device pick from the bank's
            # increments, STEPPING is the data-independent one (also what AUTO falls back to: DESIGN 3.2b)
            for form in ((SMX_FORM_AUTO, SMX_FORM_STEPPING) if carry else (SMX_FORM_AUTO,)):
                big.bank.set_block_form(form)
                reps = 50 if frames < 1024 else 5
                ms = time_saw(big, frames, reps, 5 if frames < 1024 else 1)
                if verify:
                    big.verify(frames, "saw bank %d frames" % frames, pick=sorted({0, frames // 2, frames - 1}))
                events = carry and form == SMX_FORM_AUTO
                direct = not carry or (short and form == SMX_FORM_STEPPING)
                # issue time of the stepping forms' inner loops; not defined when the wraps are located instead
                valu = None if events else ("saw_direct" if direct else "saw_carry")
                out.append(saw_entry(
                    "saw bank, %d voices, %d frames/step%s" % (voices, frames, ", form STEPPING pinned" if form == SMX_FORM_STEPPING else ""),
                    voices, frames, ms, valu,
                    {"formulation": ("carry, wrap events (AUTO: picked on the device from the bank's increments)" if events
                                     else "carry, stepping") if not direct else "direct"}))
                if events:
                    # divisions, compares, selects, multiplies: mostly the 4.3-4.7-cycle class, some 2.7-cycle adds
                    with_recorded_issue(out[-1], ("saw_bank_event_long_kernel", "1024u") if frames >= 1024
                                        else ("saw_bank_carry_kernel", "32, true" if short else "128, true" if frames > 64 else "64, true"), 4.0)
        big.bank.set_block_form(SMX_FORM_AUTO)
    if "saw_hi" in legs:
        # the same 64-frame blocks on a bank of high voices only (MIDI notes 100..127: 3..17 wraps per voice per
        # block): AUTO keeps the stepping form there, whose time does not depend on the data
        r = synthetic.splitmix64(0x5EED0009, voices)
        hi_inc = np.ascontiguousarray(tab[100 + (r % np.uint64(28)).astype(np.int64)].astype(np.uint32))
        big.bank.load(inc=hi_inc)
        big.rebase()
        ms = time_saw(big, 64, 50, 5)
        if verify:
            big.verify(64, "saw bank, high notes", pick=[0, 31, 63])
        out.append(saw_entry("saw bank, %d voices, 64 frames/step, notes 100..127 only (AUTO keeps the stepping form here)"
                             % voices, voices, 64, ms, "saw_carry"))
    if "c2" in legs:
        # BASELINE config 2: 65 536 voices, 64-frame blocks
        inc, st = synthetic.saw_bank(65536, 0x5EED0002, tab)
        b = Saw(sta, 65536, inc, st)
        ms = time_saw(b, 64, 200, 20)
        if verify:
            b.verify(64, "c2")
        out.append(saw_entry("c2: saw bank, 65536 voices, 64 frames/step (BASELINE configs[1] as written)", 65536, 64, ms, "saw_direct",
                             {"bound_note": "launch-bound: 512 KiB of state and 0.1 us of arithmetic per launch; the kernel "
                                            "floor of an empty launch is ~2-3 us"}))
        out[-1]["roofline"]["bound"] = "launch latency"
        ms = time_saw(b, 4096, 50, 5)
        if verify:
            b.verify(4096, "c2 x 4096 frames", pick=[0, 1, 63, 64, 2047, 4095])
        out.append(saw_entry("c2: saw bank, 65536 voices, 4096 frames/launch (64 JACK blocks per launch, time-parallel chunks)",
                             65536, 4096, ms, "saw_direct"))
        b.bank.close()
    if "c5" in legs:
        # BASELINE config 5's per-GPU shard: 8 Mi voices over 8 GPUs = 1 Mi voices each
        inc, st = synthetic.saw_bank(1 << 20, 0x5EED0005, tab)
        b = Saw(sta, 1 << 20, inc, st)
        for frames in (1, 64):
            ms = time_saw(b, frames, 200, 20)
            if verify:
                b.verify(frames, "c5 shard %d frames" % frames)
            out.append(saw_entry("c5 shard: saw bank, 1048576 voices (1/8 of 8 Mi), %d frame(s)/step" % frames,
                                 1 << 20, frames, ms, "saw_direct"))
        b.bank.close()
    if "c3" in legs or "c3_streams" in legs:
        # BASELINE config 3: 1 Mi PDM channels (mod_pdm.c integer path), dither 0 and seeded
        n, nt = 1 << 20, 4096
        sp, ac = synthetic.pdm_bank(n, 0x5EED0003)
        p = sta.PdmBank(n)
        p.load(sp, ac)
        dith = synthetic.dither_stream(nt, 7, 0x0FFFFFFF)          # mod_pdm.c:261 mask
        for layout in ("tick-major",) * ("c3" in legs) + ("channel-stream",) * ("c3_streams" in legs):
            run = p.tick_n_async if layout == "tick-major" else p.tick_n_streams_async
            for with_d in (False, True):
                if with_d:
                    # seeded dither for the perf leg (the generator is uc_tools': never claimed); parity tests
                    # cover explicit dither arrays
                    p.tick_n(nt, dith, want_bits=False)
                settle(lambda: run(nt, with_d), p.sync)
                p.timer_start()
                reps = 20
                for _ in range(reps):
                    run(nt, with_d)
                ms = p.timer_stop() / reps
                checked = None
                if verify:
                    _, a0 = p.read()
                    rows = [0, 1, 31, 32, nt // 2, nt - 1]
                    want = pdm_rows_closed_form(sp, a0, dith if with_d else None, rows)
                    if layout == "tick-major":
                        got = p.tick_n(nt, dith if with_d else None)
                        ok = all(np.array_equal(got[t], w) for t, w in zip(rows, want))
                    else:
                        got = p.tick_n_streams(nt, dith if with_d else None)
                        ok = all(np.array_equal((got[t // 32] >> np.uint32(t % 32)) & 1,
                                                np.unpackbits(w.view(np.uint8), bitorder="little"))
                                 for t, w in zip(rows, want))
                    if not ok:
                        sys.exit("bench.py: PDM CHECK FAILED (%s, dither=%s)" % (layout, with_d))
                    checked = "pulse rows %s == closed form" % rows
                alg = 8.0 * n + nt * n / 8.0          # setpoint + lazily materialised accumulator read, pulse bits written
                e = {"workload": "c3: carry-out PDM bank, %d channels, %d ticks/launch, %s layout, dither=%s"
                                 % (n, nt, layout, "seeded" if with_d else "0"),
                     "value": round(n * nt / (ms * 1e-3) / 1e9, 2), "unit": "Gsamples/s (channel-ticks)",
                     "ms_per_step": round(ms, 4),
                     "roofline": roof(alg, ms, n * nt, ISSUE["pdm_tick_major" if layout == "tick-major" else "pdm_stream"]
                                      + (ISSUE["dither_add"] if with_d else 0.0)),
                     "verified": checked}
                e["hbm_frac"] = e["roofline"]["hbm_frac"]
                out.append(e)
        p.close()
    if "pwm" in legs:
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
        checked = None
        if verify:
            # one more launch with a known dither stream, placed so that a control-rate boundary (line copy +
            # control_update) falls inside it: a slice of channels tick by tick against the numpy statement of the
            # firmware's channel (duty bytes and the shaper's state), and EVERY channel's glide state in closed form
            nc, first = 256, 100
            w.div_count = (1 << 12) - first
            before = w.read()
            dith = synthetic.dither_stream(nc, 11, 0x3FF)                  # mod_pdm_pwm.c:127 mask
            duty = w.tick_n(nc, dith)
            after = w.read()
            sl = np.r_[0:1024, n // 2 - 512:n // 2 + 512, n - 1024:n]
            want_duty, want = pwm2_block_numpy({k: v[sl] for k, v in before.items()}, dith, nc, (1 << 12) - first)
            ok = np.array_equal(duty[:, sl], want_duty) and all(np.array_equal(after[k][sl], want[k]) for k in want)
            with np.errstate(over="ignore"):
                pos1 = before["pos1"] + (before["vel1"] << np.uint32(12))
                ok = ok and np.array_equal(after["pos0"], before["pos1"] + np.uint32(nc - first) * before["vel1"]) \
                    and np.array_equal(after["vel0"], before["vel1"]) and np.array_equal(after["pos1"], pos1) \
                    and np.array_equal(after["vel1"], ((before["setpoint"] - pos1).view(np.int32) >> 12).view(np.uint32))
            if not ok or w.div_count != nc - first:
                sys.exit("bench.py: PWM CHECK FAILED: duty bytes / shaper state of the slice or the bank's glide state differ")
            checked = ("duty + shaper state of %d channels x %d ticks == numpy statement of mod_pdm_pwm.c's channel "
                       "(oracle-pinned shaper: pdm.h), glide state of all %d channels == closed form across a "
                       "control-rate boundary" % (len(sl), nc, n))
        w.close()
        e = {"workload": "noise-shaped PWM bank (pdm2+glide), %d channels, %d ticks/launch, dither seeded" % (n, nt),
             "value": round(n * nt / (ms * 1e-3) / 1e9, 2), "unit": "Gsamples/s (channel-ticks)",
             "ms_per_step": round(ms, 4), "roofline": roof(52.0 * n + float(nt) * n, ms, n * nt, ISSUE["pwm2"]),
             "verified": checked}
        e["hbm_frac"] = e["roofline"]["hbm_frac"]
        out.append(e)
    if "c4" in legs:
        # BASELINE config 4: 256 Ki poly voices (saw + 1-pole LPF + ADSR + stereo mix; build-defined), 64-frame blocks.
        # SURVEY 8d's scenario: T = 48 000 samples (750 blocks), every voice's gate on for the first half and off for the
        # second, envelopes at rest at t = 0, rates log-uniform 1 ms .. 1 s -- so attacks, decays and releases are in
        # progress for most of the run (the kernel's general envelope path), not only the holds of a settled bank.
        n, nblk = 1 << 18, 750
        pb = sta.PolyBank(n)
        arrs = synthetic.poly_bank(n, 0x5EED0004, tab)
        on = np.ones(n, np.uint32)
        off = np.zeros(n, np.uint32)

        def scenario(timed):
            pb.load(**dict(arrs, gate=on))
            total = 0.0
            for half, gate in enumerate((on, off)):
                if half:
                    pb.load(gate=gate)                       # the control-rate input changes between two blocks
                if timed:
                    pb.timer_start()
                for _ in range(nblk // 2):
                    pb.run_async(64)
                if timed:
                    total += pb.timer_stop()
                else:
                    pb.sync()
            return total / (2 * (nblk // 2))
        for _ in range(3):
            scenario(False)                                   # ~30 ms of untimed blocks (clock ramp: time_saw)
        ms_run = min(scenario(True) for _ in range(3))
        # the settled bank (random gates, every envelope holding): round 2's number, for continuity
        pb.load(**arrs)
        settle(lambda: pb.run_async(64), pb.sync)
        pb.timer_start()
        reps = 200
        for _ in range(reps):
            pb.run_async(64)
        ms = pb.timer_stop() / reps
        checked = None
        if verify:
            # a block in the middle of the attacks and decays (20 blocks after the gates went on) and one of the settled bank
            for what, gate, blocks in (("envelopes moving", on, 20), ("settled", arrs["gate"], 400)):
                pb.load(**dict(arrs, gate=gate))
                for _ in range(blocks):
                    pb.run_async(64)
                cur = pb.read()
                cur["gate"] = gate
                want = poly_block_numpy(cur, 64)
                got, _ = pb.run(64)
                if not np.array_equal(got.reshape(64, 2).astype(np.int64), want):
                    sys.exit("bench.py: POLY CHECK FAILED (%s): the stereo bus differs from the numpy statement of DESIGN 3.5" % what)
            checked = "stereo bus == numpy statement of the build's own definition (not reference parity), with envelopes moving and settled"
        pb.close()
        for what, t in (("the 48 000-sample run of SURVEY 8d (gates on for T/2: attacks, decays, releases in progress), mean of 750 blocks", ms_run),
                        ("settled bank (every envelope holding)", ms)):
            e = {"workload": "c4: poly bank (saw+LPF+ADSR, stereo; build-defined), %d voices, 64 frames/step, %s" % (n, what),
                 "value": round(n * 64 / (t * 1e-3) / 1e9, 2), "unit": "Gsamples/s", "ms_per_step": round(t, 5),
                 "roofline": roof(60.0 * n + 64 * 8, t, n * 64, ISSUE["poly"]), "verified": checked}
            e["hbm_frac"] = e["roofline"]["hbm_frac"]
            # 14 (down-only envelope) .. 16 (general) instructions per voice-sample; 3.9 cycles each is what a SIMD with 8 waves
            # sustains on this mix (4 Mi voices)
            with_recorded_issue(e, ("poly_bank_kernel",), 3.9)
            e["roofline"]["bound"] = ("vector issue at 4 waves per SIMD (256 Ki voices = 16 waves per CU; 14..16 dependent instructions "
                                      "per voice-sample at 5.1 cycles each, 3.9 with 8 waves per SIMD) on top of a 4.4 us one-frame "
                                      "launch; DESIGN 3.5")
            out.append(e)
    # (last: a bank of this size leaves the allocator's free space in pieces when it goes -- see the note in the leg)
    if "pdm_tick" in legs:
        # the carry-out PDM bank in the regime the %HBM metric is defined in: the tick ABI (one ISR tick per launch,
        # mod_pdm.c:177-194) on a bank that streams from HBM -- 64 Mi channels, 8 B read per channel (setpoint + the
        # lazily materialised accumulator) + 1 bit written
        n = 1 << 26
        sp, ac = synthetic.pdm_bank(n, 0x5EED0013)
        ac = (synthetic.splitmix64(0x5EED0014, n) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        p = sta.PdmBank(n)
        p.load(sp, ac)
        timed = {}
        for nt in (1, 2):
            settle(lambda: p.tick_n_async(nt, False), p.sync)
            p.timer_start()
            reps = 100
            for _ in range(reps):
                p.tick_n_async(nt, False)
            timed[nt] = p.timer_stop() / reps
        for nt in (1, 2):
            ms = timed[nt]
            checked = None
            if verify:
                _, a0 = p.read()
                got = p.tick_n(nt)
                want = pdm_rows_closed_form(sp, a0, None, list(range(nt)))
                if not all(np.array_equal(got[t], w) for t, w in zip(range(nt), want)):
                    sys.exit("bench.py: PDM CHECK FAILED (tick ABI, %d tick(s))" % nt)
                checked = "all %d pulse row(s) == closed form" % nt
            alg = 8.0 * n + nt * n / 8.0
            e = {"workload": "carry-out PDM bank, %d channels, %d tick(s)/launch (tick ABI), dither=0" % (n, nt),
                 "value": round(n * nt / (ms * 1e-3) / 1e9, 2), "unit": "Gsamples/s (channel-ticks)",
                 "ms_per_step": round(ms, 5), "roofline": roof(alg, ms), "verified": checked}
            e["hbm_frac"] = e["roofline"]["hbm_frac"]
            out.append(e)
        p.close()
        # Order matters for what FOLLOWS a leg like this one: after its 0.5 GB arrays and their read-backs are gone,
        # the 1 GiB duty buffer of the PWM leg came to lie in recycled memory and its tick-major rows (1 MB apart, 1024
        # of them written side by side) ran at 0.82 ms instead of 0.21 ms per launch, steadily -- not a transient that
        # settling cures.  So this leg runs after the others.
    return out


# ---------------------------------------------------------------------------------------------
# N > 1: the legs that only a run over several GPUs can measure (every rank makes the same calls: SPMD)
# ---------------------------------------------------------------------------------------------
def gather_floats(rdzv, values):
    """-> list over ranks of the ranks' float lists."""
    import struct
    parts = rdzv.allgather(struct.pack("<%dd" % len(values), *values))
    return [list(struct.unpack("<%dd" % len(values), p)) for p in parts]


def multi_gpu_legs(sta, synthetic, tab, rdzv, rank, world, local, verify, L):
    """BASELINE config 5 AS WRITTEN on the communicator: 1 Mi voices per GPU (8 Mi over 8 GPUs), per-GPU int32 mix,
    RCCL sum (linux/synth.c:172-179 sharded).
      (i)   throughput mode (run_async + allreduce_async) at 1 and 64 frames per step with 1 / 8 / 16 blocks per
            collective: Gsamples/s over all ranks, per-GPU %HBM, collectives issued;
      (ii)  the same shard through the synchronous and the pipelined smx_bank_run: wall microseconds per block;
      (iii) collectives alone (256 B / 2 KB / 32 KB sums, no kernel in between): the measured latency L that
            DESIGN 4's budget had to assume;
    every timed loop bracketed by barrier + device synchronize on both sides, MAX over ranks; every leg's bus
    (summed over ranks) checked against the closed form.  Returns (entries, collective probe) on every rank."""
    V = 1 << 20
    inc, st = synthetic.saw_bank(V, 0x5EED0C05 + 0x1000 * rank, tab)
    c5 = Saw(sta, V, inc, st, device=local)
    bank = c5.bank
    uid = rdzv.broadcast(sta.comm_unique_id().tobytes() if rank == 0 else b"")
    bank.comm_init(rank, world, np.frombuffer(uid, np.uint8).copy())
    out = []

    def bracket(steps, body):
        """-> (max over ranks of the wall seconds, per-rank HIP-event ms per step)"""
        rdzv.barrier()
        L.smx_device_synchronize(local)
        t0 = time.perf_counter()
        bank.timer_start()
        for _ in range(steps):
            body()
        kms = bank.timer_stop() / steps
        bank.sync()
        L.smx_device_synchronize(local)
        rdzv.barrier()
        dt = time.perf_counter() - t0
        per_rank = gather_floats(rdzv, [dt, kms])
        return max(p[0] for p in per_rank), [p[1] for p in per_rank]

    def check_run(frames, pipelined):
        """smx_bank_run's own return value (sum over ranks; one block late when pipelined) == closed form."""
        t_first = c5.t
        bus, _ = bank.run(frames)
        c5.t += frames
        if pipelined:
            bus, _ = bank.run(frames)                  # hands out the block launched by the call before
            c5.t += frames
        want = saw_bus_closed_form(c5.inc, c5.state0, t_first, list(range(frames)))
        parts = rdzv.allgather(want.astype("<i8").tobytes())
        want = wrap_i32(sum(np.frombuffer(p, "<i8") for p in parts))
        if not rdzv.all_ok(bool(np.array_equal(bus.astype(np.int64), want))):
            sys.exit("bench.py: BUS CHECK FAILED (c5 shard, smx_bank_run, %d frames, %s)"
                     % (frames, "pipelined" if pipelined else "sync"))

    # (i) throughput mode
    for frames in (1, 64):
        steps = 400 if frames == 1 else 200
        for group in (1, 8, 16):
            bank.set_comm_group(group)

            def step():
                c5.run_async(frames)
                bank.allreduce_async(frames)
            for _ in range(20):
                step()
            bank.sync()
            c0 = bank.comm_stats()
            dt, kms = bracket(steps, step)
            c1 = bank.comm_stats()
            if verify:
                c5.verify(frames, "c5 as written, %d frames, group %d" % (frames, group), rdzv=rdzv, comm=True,
                          pick=None if frames <= 8 else [0, frames // 2, frames - 1])
            ms = dt / steps * 1e3
            e = saw_entry("c5 as written: %d x 1048576 voices (1 Mi per GPU), %d frame(s)/step, throughput mode, "
                          "%d block(s) per collective" % (world, frames, group), V, frames, ms, "saw_direct",
                          {"n_gpus": world, "comm_group": group,
                           "collectives_issued": int(c1[0] - c0[0]), "block_sums_carried": int(c1[1] - c0[1]),
                           "kernel_ms_per_rank": {"min": round(min(kms), 5), "max": round(max(kms), 5)},
                           "timing": "wall, barrier + device sync on both sides, max over ranks"})
            e["value"] = round(world * V * frames / (ms * 1e-3) / 1e9, 2)         # whole job
            e["max_voices_48k"] = int(world * V * frames / (ms * 1e-3) / 48000)
            e["verified"] = "bus summed over %d ranks == closed form" % world if verify else None
            out.append(e)
    bank.set_comm_group(8)
    # (ii) smx_bank_run as the JACK callback calls it: synchronous, then pipelined
    for pipelined in (False, True):
        bank.set_block_mode(pipelined)
        for frames in (1, 64):
            for _ in range(10):
                bank.run(frames)
            c5.t += 10 * frames
            steps = 200

            def block():
                bank.run(frames)
            dt, kms = bracket(steps, block)
            c5.t += steps * frames
            if verify:
                check_run(frames, pipelined)
            out.append({"workload": "c5 shard through smx_bank_run (%s), %d x 1048576 voices, %d frame(s)/block"
                                    % ("pipelined: block k-1 handed out while k runs" if pipelined else "synchronous", world, frames),
                        "n_gpus": world, "us_per_block": round(dt / steps * 1e6, 2), "unit": "us per block (wall, max over ranks)",
                        "value": round(world * V * frames * steps / dt / 1e9, 2), "value_unit": "Gsamples/s",
                        "verified": "returned bus (sum over %d ranks) == closed form" % world if verify else None})
    bank.set_block_mode(False)
    # (iii) the collective alone
    probe = []
    for words in (64, 512, 8192):
        us_sync, us_queued = bank.comm_probe(words, 100)
        per_rank = gather_floats(rdzv, [us_sync, us_queued])
        probe.append({"bytes": words * 4, "int32_words": words,
                      "collective_us": round(max(p[0] for p in per_rank), 2),
                      "collective_us_queued": round(max(p[1] for p in per_rank), 2)})
    rdzv.barrier()
    bank.close()
    return out, probe


# ---------------------------------------------------------------------------------------------
def run_rank(a):
    # stdout carries exactly ONE line (the JSON): libraries that print banners at init (RCCL
    # prints its version block to stdout) are pointed at stderr for the whole run.
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    a.gpus = world

    import synth_tools_amd as sta
    from synth_tools_amd import rendezvous, synthetic

    L = sta.lib()
    ndev = L.smx_device_count()
    if os.environ.get("SMX_BENCH_DEVICES"):
        # rehearsal on fewer GPUs than ranks (tests/test_multi_rank_gpu.py with its RCCL test double): "0,0" = both
        # ranks on device 0.  Real RCCL refuses two ranks on one device.
        local = int(os.environ["SMX_BENCH_DEVICES"].split(",")[local])
    if ndev <= local:
        sys.exit("bench.py: rank %d: no GPU visible for local rank %d (%d device(s); there is no CPU fallback)"
                 % (rank, local, ndev))
    rdzv = rendezvous.Rendezvous(rank, world)

    tab = synthetic.note_inc_table(L.note_to_inc)
    # rank r owns voices [r*V, (r+1)*V) of the global bank: its own splitmix64 stream
    inc, state = synthetic.saw_bank(a.voices, 0x5EED0005 + 0x1000 * rank, tab)
    saw = Saw(sta, a.voices, inc, state, device=local)
    bank = saw.bank

    comm = world > 1
    if comm:
        uid = rdzv.broadcast(sta.comm_unique_id().tobytes() if rank == 0 else b"")
        bank.comm_init(rank, world, np.frombuffer(uid, np.uint8).copy())
    elif os.environ.get("SMX_BENCH_FORCE_COMM"):
        # rehearsal of the multi-GPU code path on one GPU: 1-rank RCCL communicator
        bank.comm_init(0, 1, sta.comm_unique_id())
        comm = True
    ranks_seen = bank.comm_ranks() if comm else 1
    for _ in range(a.warmup):
        saw.run_async(a.frames)
        if comm:
            bank.allreduce_async(a.frames)
    bank.sync()

    # The timed region: EXACTLY --steps steps between barrier + device synchronize on both sides, max over ranks --
    # run --repeats times back to back; the line reports the MEDIAN region (and the fastest and slowest beside it):
    # one 1.6 ms sample on a pool whose boxes and processes differ by 1-7 % is fragile.
    regions = []
    for _ in range(max(1, a.repeats)):
        rdzv.barrier()
        L.smx_device_synchronize(local)
        t0 = time.perf_counter()
        bank.timer_start()
        for _ in range(a.steps):
            saw.run_async(a.frames)
            if comm:
                bank.allreduce_async(a.frames)
        k_ms = bank.timer_stop() / a.steps         # HIP events on the kernel's stream
        bank.sync()                                # issues what is still queued, both streams idle
        L.smx_device_synchronize(local)
        rdzv.barrier()
        d = time.perf_counter() - t0
        per_rank = gather_floats(rdzv, [d, k_ms])
        regions.append((max(p[0] for p in per_rank), max(p[1] for p in per_rank), [p[1] for p in per_rank]))
    regions.sort(key=lambda r: r[0])
    dt, kernel_ms, kernel_ms_ranks = regions[len(regions) // 2]
    dt_min, dt_max = regions[0][0], regions[-1][0]

    # the timed kernel's only output is the bus: one more step, compared with the closed form
    checked = 0
    if not a.no_verify:
        checked = saw.verify(a.frames, "headline, %d voices x %d frames" % (a.voices, a.frames), rdzv=rdzv, comm=comm,
                             pick=None if a.frames <= 8 else sorted({0, a.frames // 2, a.frames - 1}))
    collectives, block_sums = bank.comm_stats() if comm else (0, 0)
    mgpu = None
    if comm and not a.no_also:
        mgpu = multi_gpu_legs(sta, synthetic, tab, rdzv, rank, world, local, not a.no_verify, L)

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
            "ms_per_step_min": round(dt_min / a.steps * 1e3, 5), "ms_per_step_max": round(dt_max / a.steps * 1e3, 5),
            "repeats": len(regions),
            "timing": "median of %d timed regions of %d steps each (barrier + device synchronize on both sides of "
                      "every region, max over ranks); value and ms_per_step are the median region's" % (len(regions), a.steps),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "saw voice bank resident in HBM, %d voices/GPU x %d frame(s)/step "
                                   "(%s), seeded note bank (splitmix64), all voices on"
                                   % (a.voices, a.frames, "tick ABI" if a.frames == 1 else "process() blocks"),
                       "voices_per_gpu": a.voices, "frames_per_step": a.frames,
                       "voices_total": world * a.voices,
                       "baseline_config": "BASELINE configs[1] (int32 phase-accumulator saw voices on 1 MI355X, "
                                          "bit-exact) scaled from 65 536 voices (512 KiB, launch-bound) to an "
                                          "HBM-resident bank, the regime the %HBM-roofline metric is defined in; every "
                                          "BASELINE config as written is a line of `also` with its own roofline block",
                       "parallelism": "voices sharded x%d, int32 bus all-reduce (RCCL), one collective per %d steps"
                                      % (world, max(1, block_sums // max(1, collectives))) if comm else "1 GPU"},
            "max_voices_48k": int(vs / 48000),
            "ranks_seen": ranks_seen,
            "multi_gpu_measured": "this line" if world > 1 else "no N > 1 run is part of this line",
            "verified": ("bus of one more step == closed form sum_v ((int32)(state0 + T*inc) >> 4), %d frame(s), %d voices%s"
                         % (checked, world * a.voices, " summed over %d ranks" % world if world > 1 else ""))
                        if checked else None,
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": "recorded: profiles/traffic.json (rocprofv3 --pmc passes of this command), "
                                           "not measured in this run" if traffic else None,
                         "kernel": "saw_tick_kernel" if (a.frames <= 4 and a.voices >= 1 << 20 and a.voices % 4096 == 0)
                                   else "saw_bank_kernel",
                         "kernel_ms": round(kernel_ms, 5),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "bytes_per_voice": 8,
                         "note": "8 B read per voice per launch (inc + phase base); the advanced phase is kept as "
                                 "state0 + elapsed*inc and never written back, so SURVEY 8d's 4-byte state write "
                                 "(12 B/voice) does not exist on this path; PMC traffic agrees (profiles/)"},
        }
        if comm:
            line["collectives"] = {"issued": collectives, "block_sums_carried": block_sums}
            line["roofline"]["kernel_ms_per_rank"] = {"min": round(min(kernel_ms_ranks), 5), "max": round(max(kernel_ms_ranks), 5)}
        if mgpu is not None:
            # N > 1 (or a 1-rank communicator forced for a rehearsal): BASELINE config 5 as written + the collective alone
            line["also"] = mgpu[0]
            line["collective_probe"] = {"what": "ncclAllReduce(int32, sum) of a bus of that size on the bank's communicator, "
                                                "no kernel in between, 100 repetitions, max over ranks: collective_us = each one "
                                                "waited for (launch + collective + stream sync), collective_us_queued = back to "
                                                "back (the comm stream's own rate)",
                                        "sizes": mgpu[1]}
        if world == 1 and not a.no_cpu:
            ref = cpu_baseline_reference(inc, state, min(a.cpu_seconds, 4.0))
            single, par = cpu_baseline_saw(inc, state, a.frames, a.cpu_seconds)
            line["cpu_baseline"] = ref if ref else single
            line["cpu_baseline_port"] = single
            line["cpu_baseline_parallel"] = par
            line["cpu_baselines_extra"] = cpu_baselines_extra(synthetic, tab)
        if world == 1 and not a.no_also:
            legs = [x for x in a.legs.split(",") if x]
            also = also_workloads(sta, synthetic, tab, saw, a.voices, legs, not a.no_verify)
            line["also"] = also + (mgpu[0] if mgpu is not None else [])
            cfg = [e for e in also if e["workload"].startswith(("c2:", "c3:", "c4:", "c5 "))]
            line["config"]["baseline_configs_vs_50pct_hbm"] = {
                "meet": [e["workload"] for e in cfg if e["roofline"]["hbm_frac"] >= 0.5],
                "miss": [{"workload": e["workload"], "hbm_frac": e["roofline"]["hbm_frac"], "bound": e["roofline"]["bound"]}
                         for e in cfg if e["roofline"]["hbm_frac"] < 0.5]}
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    rdzv.barrier()              # every rank is done with its communicator before any is torn down
    bank.close()
    rdzv.close()


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus))
    run_rank(a)


if __name__ == "__main__":
    main()
