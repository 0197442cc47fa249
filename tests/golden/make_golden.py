"""Regenerates tests/golden/*.  Run in the build container (needs /root/reference
for the pdm.h vectors): `python tests/golden/make_golden.py`.

 survey_known_answers.json  hand-entered DATA: the known-answers SURVEY.md
                            (§8 a-5, Appendix A.2/A.3) recorded from the
                            reference, plus the one KAT the reference holds
                            (comment at stm32f103/mod_pdm.c:30-53: the power-of-two period
                            table, the X=3 rows and the X=5 accumulator row).
 pdm_h_reference.npz        outputs of the REAL stm32f103/pdm.h (oracle/_ref,
                            compiled from /root/reference where it lies).
 synth_run_derived.npz      regression vectors of THIS repo's restatement of
                            linux/synth.c (derived, not reference output).
 synth_c_reference.npz      outputs of the REAL linux/synth.c:27-208 (oracle/_ref/libref_synth.so,
                            compiled verbatim from /root/reference): note_to_inc(0..127), midi_tab,
                            float bits + final voice[] + note2voice[] of scripted sequences.
 pmeas_reference.npz        struct pmeas_state after every call of the REAL pmeas_update
                            (stm32f103/pmeas.h:64-108, oracle/_ref/libref_pmeas.so).
 pwmosc_reference.npz       duty bytes and phases of the REAL pwm_update / OSC_HARD_SYNC
                            (stm32f103/mod_pdm.c:159-175, oracle/_ref/libref_pwmosc.so): the reference's
                            default speed 256*13 and five others, 70 000 ticks each, hard syncs at
                            scripted ticks.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402
from synth_tools_amd import synthetic  # noqa: E402


def survey_known_answers():
    d = {
        "_provenance": "SURVEY.md §8 a-5 and Appendix A.2/A.3 (values observed from the reference "
                       "during the survey) and the comment KATs at stm32f103/mod_pdm.c:30-38 (period table), "
                       ":43-47 (X=3 rows), :49-53 (X=5 accumulator row)",
        "note_tab": [594573364, 629928536, 667386036, 707070875, 749115497, 793660223,
                     840853716, 890853479, 943826384, 999949221, 1059409296, 1122405051],
        "note_to_inc": {"0": 731558, "16": 1843410, "32": 4645104, "48": 11704929,
                        "64": 29494574, "69": 39370533, "80": 74321670, "96": 187278874,
                        "112": 471913192},
        "note_to_inc_sum_0_127": 19985760579,
        "chord_69_72_76_first8_float_bits": ["00000000", "3b0a742f", "3b8a7430", "3bcfae48",
                                              "3c0a7430", "3c2d113c", "3c4fae48", "3c724b54"],
        "chord_voice0_after8": {"inc": 0x0258BF25, "state": 0x12C5F928},
        "pdm_h_in2000000000_sh24_16calls": {"sum_pdm1": 1788, "sum_pdm2": 1789, "sum_pdm3": 1788,
                                             "pdm1_s1": 0x77594000,
                                             "pdm2_s": [0x76594000, 0x7676A000]},
        "mod_pdm_comment_kat_3bit": {"X": 3, "A": [0, 3, 6, 1, 4, 7, 2, 5, 0],
                                     "C": [1, 0, 0, 1, 0, 0, 1, 0, 1]},
        # the complementary row of the same comment (mod_pdm.c:49-53, X = 5 = 8 - 3).  Only its A row is data: the
        # C row printed under it (1 0 1 0 0 1 0 0 1) is NOT what add-with-carry gives (that is 1 0 1 0 1 1 0 1 1;
        # the comment mirrors the X = 3 pulses instead) -- SURVEY §4 -- and is kept here only to say so.
        "mod_pdm_comment_kat_3bit_x5": {"X": 5, "A": [0, 5, 2, 7, 4, 1, 6, 3, 0],
                                        "C_as_printed_not_reproducible": [1, 0, 1, 0, 0, 1, 0, 0, 1]},
        # the power-of-two table of the same comment (mod_pdm.c:30-38): resolution N = 2^B, input X a power of two
        # -> one pulse every P samples.  Rows as [X, P] in units of N (X = N/d -> P = d); the last row (X = N,
        # P = 1) is outside a B-bit input and is listed for completeness only.
        "mod_pdm_comment_period_table": {"rows_X_over_N__P": [["1/N", "N"], ["2/N", "N/2"], ["4/N", "N/4"],
                                                                ["1/2", 2], ["1", 1]],
                                         "testable_log2_X_32bit": [20, 21, 22, 24, 27, 29, 30, 31]},
        "mod_pdm_two_channel_derived": {"setpoint": [2000000000, 0x40000000], "ticks": 6,
                                        "bsrr": [0x300000, 0x300000, 0x200010, 0x100020,
                                                 0x200010, 0x300000],
                                        "accu_end": [0xCB417800, 0x80000000]},
    }
    with open(os.path.join(HERE, "survey_known_answers.json"), "w") as f:
        json.dump(d, f, indent=1)


def pdm_h_reference():
    ref = oracle.load_ref_pdm()
    if ref is None:
        print("oracle/_ref/libref_pdm.so absent and /root/reference not present: skipped")
        return
    steps = 192
    inputs = np.array([2000000000, 0x40000000, 0xC0000000, 12345, 0xFFFFFFFF, 0x80000001], np.uint32)
    shifts = np.array([24, 16, 31, 1], np.uint32)
    dith = {0: np.zeros(steps, np.uint32),
            1: synthetic.dither_stream(steps, 0x5EED0D17, 0x3FF)}      # mod_pdm_pwm.c:127 mask
    out = {"inputs": inputs, "shifts": shifts, "dither1": dith[1], "steps": np.uint32(steps)}
    for order in (1, 2, 3, 4):
        f = getattr(ref, "ref_pdm%d_update" % order)
        q = np.zeros((len(inputs), len(shifts), 2, steps), np.uint32)
        fin = np.zeros((len(inputs), len(shifts), 2, order), np.uint32)
        for i, x in enumerate(inputs):
            for j, sh in enumerate(shifts):
                for k in (0, 1):
                    s = np.zeros(order, np.uint32)
                    for t in range(steps):
                        if order == 1:
                            q[i, j, k, t] = f(s, int(x), int(sh))
                        else:
                            q[i, j, k, t] = f(s, int(x), int(sh), int(dith[k][t]))
                    fin[i, j, k] = s
        out["q%d" % order] = q
        out["s%d" % order] = fin
    np.savez_compressed(os.path.join(HERE, "pdm_h_reference.npz"), **out)


def synth_run_derived():
    lib = oracle.load()
    n2v = np.zeros(128, np.int32)
    inc = np.zeros(64, np.uint32)
    st = np.zeros(64, np.uint32)
    script = [("on", 60), ("run", 64), ("on", 64), ("on", 67), ("run", 64), ("off", 64),
              ("run", 1), ("off", 99), ("run", 64)]                      # stray note-off
    script += [("on", n) for n in range(20, 90)] + [("run", 256)]        # > 64 notes: steals voice 0
    script += [("off", n) for n in range(20, 90, 3)] + [("run", 64)]
    vecs = []
    for op, a in script:
        if op == "on":
            lib.orc_note_on(n2v, inc, 64, a)
        elif op == "off":
            lib.orc_note_off(n2v, inc, 64, a)
        else:
            _, v = oracle.synth_run(lib, inc, st, a)
            vecs.append(v)
    np.savez_compressed(os.path.join(HERE, "synth_run_derived.npz"),
                        script=np.array([[0 if o == "on" else 1 if o == "off" else 2, a] for o, a in script], np.int32),
                        vec=np.concatenate(vecs), inc=inc, state=st, note2voice=n2v)


# ---- linux/synth.c:27-208 and pmeas.h:64-108, compiled VERBATIM into oracle/_ref ------------
# op codes of a script row [op, a, b]
OP_ON, OP_OFF, OP_RUN, OP_SQUARE, OP_POKE = 0, 1, 2, 3, 4


def synth_scripts():
    """Scripted note-on/off/run sequences: allocator quirks (steal voice 0
    when full, stray note-off silences voice 0, note % 128, phase not reset by note_on),
    >= 16 full-scale voices so the reference's `int sum` wraps, B in {1, 64, 4096},
    sum_tick_square."""
    rng = np.random.default_rng(0x5EED0C)
    s = {}
    s["b1_ticks"] = ([(OP_ON, 69, 0)] + [(OP_RUN, 1, 0)] * 5 + [(OP_ON, 72, 0), (OP_RUN, 1, 0),
                     (OP_ON, 76, 0)] + [(OP_RUN, 1, 0)] * 8 + [(OP_OFF, 69, 0)] + [(OP_RUN, 1, 0)] * 4 +
                     [(OP_OFF, 5, 0), (OP_RUN, 1, 0), (OP_ON, 69 + 128, 0), (OP_RUN, 1, 0),
                      (OP_OFF, 72 + 256, 0)] + [(OP_RUN, 1, 0)] * 3)
    q = [(OP_ON, 60, 0), (OP_RUN, 64, 0), (OP_ON, 64, 0), (OP_ON, 67, 0), (OP_RUN, 64, 0), (OP_OFF, 64, 0),
         (OP_RUN, 1, 0), (OP_OFF, 99, 0), (OP_RUN, 64, 0)]                      # stray note-off
    q += [(OP_ON, n, 0) for n in range(20, 90)] + [(OP_RUN, 256, 0)]            # > 64 notes: steals voice 0
    q += [(OP_OFF, n, 0) for n in range(20, 90, 3)] + [(OP_RUN, 64, 0)]
    q += [(OP_ON, 127, 0), (OP_ON, 0, 0), (OP_ON, 127, 0), (OP_RUN, 64, 0), (OP_OFF, 127, 0), (OP_RUN, 64, 0)]
    s["b64_quirks"] = q
    q = [(OP_ON, int(n), 0) for n in rng.integers(0, 128, 40)] + [(OP_RUN, 4096, 0)]
    q += [(OP_OFF, int(n), 0) for n in rng.integers(0, 128, 30)] + [(OP_RUN, 4096, 0)]
    q += [(OP_ON, int(n), 0) for n in rng.integers(30, 100, 50)] + [(OP_RUN, 4096, 0)]
    s["b4096_random"] = q
    # 64 voices sounding, phases near +full scale / -full scale: the int sum wraps (linux/synth.c:171-176)
    q = [(OP_ON, n, 0) for n in range(64, 128)]
    q += [(OP_POKE, v, 0x7FFFFF00 - 977 * v) for v in range(64)] + [(OP_RUN, 64, 0)]
    q += [(OP_POKE, v, 0x80000000 + 4099 * v) for v in range(64)] + [(OP_RUN, 64, 0)]
    q += [(OP_POKE, v, int(x)) for v, x in enumerate(rng.integers(0, 2**32, 64))] + [(OP_RUN, 1, 0), (OP_RUN, 64, 0)]
    s["wrapping_mix"] = q
    q = [(OP_ON, 40, 0), (OP_SQUARE, 300, 0), (OP_ON, 47, 0), (OP_ON, 52, 0), (OP_SQUARE, 300, 0),
         (OP_OFF, 40, 0), (OP_SQUARE, 64, 0), (OP_RUN, 64, 0), (OP_OFF, 47, 0), (OP_OFF, 52, 0), (OP_SQUARE, 8, 0)]
    s["square"] = q
    return {k: np.array(v, np.int64) for k, v in s.items()}


def run_script_on_reference(ref, script):
    import ctypes as C
    x = oracle.RefSynth()
    ref.synth_init(C.byref(x))
    out = []
    with oracle.quiet_stderr():
        for op, a, b in script:
            a = int(a)
            if op == OP_ON:
                ref.synth_note_on(C.byref(x), a)
            elif op == OP_OFF:
                ref.synth_note_off(C.byref(x), a)
            elif op == OP_RUN:
                v = np.zeros(a, np.float32)
                ref.synth_run(C.byref(x), v, a)
                out.append(v)
            elif op == OP_SQUARE:
                out.append(np.array([ref.sum_tick_square(C.byref(x)) for _ in range(a)], np.float32))
            elif op == OP_POKE:
                x.voice[a].note_state = int(b)
    n2v, inc, st = x.arrays()
    return np.concatenate(out), n2v, inc, st


def synth_c_reference():
    ref = oracle.load_ref_synth()
    if ref is None:
        print("oracle/_ref/libref_synth.so absent and /root/reference not present: skipped")
        return
    out = {"_provenance": np.array("outputs of /root/reference/linux/synth.c:27-208 compiled verbatim "
                                   "(oracle/Makefile ref) with gcc -O2 -fwrapv; generator tests/golden/make_golden.py")}
    with oracle.quiet_stderr():
        out["note_to_inc"] = np.array([ref.note_to_inc(n) for n in range(128)], np.uint32)
    out["midi_tab"] = np.array(ref.ref_midi_tab[:], np.uint8)
    for name, script in synth_scripts().items():
        vec, n2v, inc, st = run_script_on_reference(ref, script)
        out[name + "_script"] = script
        out[name + "_vec_bits"] = vec.view(np.uint32)
        out[name + "_note2voice"] = n2v
        out[name + "_inc"] = inc
        out[name + "_state"] = st
    np.savez_compressed(os.path.join(HERE, "synth_c_reference.npz"), **out)


def pmeas_traces():
    """Timestamp series for pmeas_update: steady, jittered, periods longer than the
    window, a 32-bit cycle-counter wrap, several log_max."""
    rng = np.random.default_rng(0x5EED0E)
    t = {}
    t["steady_1000_lm14"] = (14, np.cumsum(np.full(200, 1000, np.uint64)))
    t["jitter_lm26"] = (26, np.cumsum(rng.integers(160000, 170000, 1500).astype(np.uint64)))      # ~440 Hz @ 72 MHz
    t["slow_lm10"] = (10, np.cumsum(rng.integers(500, 3000, 300).astype(np.uint64)))             # period > window
    t["wrap_lm20"] = (20, 0xFFF00000 + np.cumsum(rng.integers(1000, 90000, 400).astype(np.uint64)))
    t["mixed_lm16"] = (16, np.cumsum(np.concatenate([rng.integers(1, 50, 300), rng.integers(20000, 70000, 50),
                                                     rng.integers(1, 5000, 300)]).astype(np.uint64)))
    t["lm30"] = (30, np.cumsum(rng.integers(2**24, 2**28, 300).astype(np.uint64)))
    return {k: (lm, (cc & 0xFFFFFFFF).astype(np.uint32)) for k, (lm, cc) in t.items()}


def pmeas_reference():
    ref = oracle.load_ref_pmeas()
    if ref is None:
        print("oracle/_ref/libref_pmeas.so absent and /root/reference not present: skipped")
        return
    out = {"_provenance": np.array("state of struct pmeas_state after every pmeas_update call of "
                                   "/root/reference/stm32f103/pmeas.h:64-108 compiled verbatim (oracle/Makefile ref)"),
           "fields": np.array(oracle.RefPmeas.FIELDS)}
    for name, (lm, cc) in pmeas_traces().items():
        p = oracle.RefPmeas(ref, lm)
        snap = np.zeros((len(cc), len(oracle.RefPmeas.FIELDS)), np.uint32)
        for i, c in enumerate(cc):
            p.update(c)
            snap[i] = p.snapshot()
        out[name + "_log_max"] = np.uint32(lm)
        out[name + "_cc"] = cc
        out[name + "_trace"] = snap
    np.savez_compressed(os.path.join(HERE, "pmeas_reference.npz"), **out)


def pwmosc_cases():
    """(phase0, speed, sync ticks) per case.  70 000 ticks: the `phase >> 9` feedback makes the ramp
    accelerate, so the 24-bit mask (mod_pdm.c:162) is crossed tens to hundreds of times per case."""
    rng = np.random.default_rng(0x5EED0F)
    nt = 70000
    c = {}
    c["default_256x13"] = (0, None, np.array([], np.int64))              # speed = the reference's own initialiser
    c["default_synced"] = (0, None, np.sort(rng.choice(nt, 40, replace=False)))
    c["speed_1"] = (0, 1, np.array([5, 6, 7, 30000], np.int64))          # slowest: feedback dominates late
    c["speed_65536_phase_mid"] = (0x7FFFFF, 65536, np.sort(rng.choice(nt, 200, replace=False)))
    c["speed_max24"] = (0xFFFFFF, 0xFFFFFF, np.sort(rng.choice(nt, 25, replace=False)))
    c["speed_big_u32"] = (0x123456, 0xDEADBEEF, np.arange(0, nt, 4097))  # speed above the mask: wraps mod 2^32 first
    return nt, c


def pwmosc_reference():
    ref = oracle.load_ref_pwmosc()
    if ref is None:
        print("oracle/_ref/libref_pwmosc.so absent and /root/reference not present: skipped")
        return
    nt, cases = pwmosc_cases()
    out = {"_provenance": np.array("duty = pwm_update() and pwm_phase after every tick of "
                                   "/root/reference/stm32f103/mod_pdm.c:159-175 compiled verbatim (oracle/Makefile ref); "
                                   "sync[t]: OSC_HARD_SYNC() before tick t"),
           "nticks": np.uint32(nt), "default_speed": np.uint32(ref.ref_pwm_get_speed()),
           "default_phase": np.uint32(ref.ref_pwm_get_phase()), "control_div": np.uint32(ref.ref_pwm_control_div())}
    for name, (phase0, speed, sync_ticks) in cases.items():
        if speed is None:
            speed = int(out["default_speed"])
        ref.ref_pwm_set(int(phase0), int(speed))
        sync = np.zeros(nt, np.uint8)
        sync[sync_ticks] = 1
        duty = np.zeros(nt, np.uint8)
        ph = np.zeros(nt, np.uint32)
        ref.ref_pwmosc_run(nt, sync.ctypes.data, duty.ctypes.data, ph.ctypes.data)
        out[name + "_phase0"] = np.uint32(phase0)
        out[name + "_speed"] = np.uint32(speed)
        out[name + "_sync_ticks"] = np.asarray(sync_ticks, np.int64)
        out[name + "_duty"] = duty
        out[name + "_phase_every16"] = ph[15::16].copy()          # pwm_phase after ticks 15, 31, ... (fixture size)
        out[name + "_phase_end"] = np.uint32(ph[-1])
        out[name + "_wraps"] = np.uint32(np.count_nonzero(ph[1:] < ph[:-1]))   # crossings of the 24-bit mask + syncs
    np.savez_compressed(os.path.join(HERE, "pwmosc_reference.npz"), **out)


if __name__ == "__main__":
    survey_known_answers()
    pdm_h_reference()
    synth_run_derived()
    synth_c_reference()
    pmeas_reference()
    pwmosc_reference()
    print(sorted(os.listdir(HERE)))
