"""Regenerates tests/golden/*.  Run in the build container (needs /root/reference
for the pdm.h vectors): `python tests/golden/make_golden.py`.

 survey_known_answers.json  hand-entered DATA: the known-answers SURVEY.md
                            (§8 a-5, Appendix A.2/A.3) recorded from the
                            reference, plus the one KAT the reference holds
                            (comment at stm32f103/mod_pdm.c:43-47, X=3 row).
 pdm_h_reference.npz        outputs of the REAL stm32f103/pdm.h (oracle/_ref,
                            compiled from /root/reference where it lies).
 synth_run_derived.npz      regression vectors of THIS repo's restatement of
                            linux/synth.c (derived, not reference output).
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
                       "during the survey) and the comment KAT at stm32f103/mod_pdm.c:43-47",
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


if __name__ == "__main__":
    survey_known_answers()
    pdm_h_reference()
    synth_run_derived()
    print(sorted(os.listdir(HERE)))
