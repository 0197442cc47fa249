"""tools/summarize_counters.py <tag> -- fold gpurun_out/prof_<tag>_counters/ (tools/prof_counters.sh) into
profiles/<tag>_counters.json: per kernel, averages over its launches of the SQ instruction / busy counters and the
GRBM clock counter, with the ratios derived from them:
  valu_insts_per_wave      SQ_INSTS_VALU / waves (waves from SQ_WAVES is not collected: per launch from the grid is not
                           known here, so instruction counts are reported per launch)
  clock_GHz                GRBM_GUI_ACTIVE / 8 XCDs / kernel time (MI355X_MICROARCH.md, DVFS give-back; reads high on
                           dispatches shorter than ~0.3 ms)
  valu_active_share        SQ_ACTIVE_INST_VALU / SQ_ACTIVE_INST_ANY+SQ_WAIT_INST_ANY+SQ_WAIT_ANY (= share of the waves'
                           lifetime spent issuing vector instructions; the three terms are disjoint and sum to
                           SQ_WAVE_CYCLES, same guide)
The calibration rows (tools/ubench/valu_rates kernels, issue-bound by construction) show what these ratios read on
a kernel that does nothing but issue vector instructions."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "prof_%s_counters" % tag)


def load(leg):
    f = max(glob.glob(os.path.join(src, leg, "*", "*counter_collection.csv")), key=os.path.getmtime)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def durations():
    fs = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
    if not fs:
        return {}
    return {r["Name"]: float(r["AverageNs"]) for r in csv.DictReader(open(max(fs, key=os.path.getmtime)))}


def rows(agg, dur):
    out = []
    for k, c in agg.items():
        if "rocclr" in k:
            continue
        m = {n: sum(v) / len(v) for n, v in c.items()}
        r = {"kernel": k[:110], "launches": len(next(iter(c.values())))}
        r.update({n: round(v, 1) for n, v in m.items()})
        life = m.get("SQ_ACTIVE_INST_ANY", 0) + m.get("SQ_WAIT_INST_ANY", 0) + m.get("SQ_WAIT_ANY", 0)
        if life:
            r["valu_active_share_of_wave_lifetime"] = round(m.get("SQ_ACTIVE_INST_VALU", 0) / life, 4)
            r["any_active_share_of_wave_lifetime"] = round(m.get("SQ_ACTIVE_INST_ANY", 0) / life, 4)
            r["issue_stall_share_of_wave_lifetime"] = round(m.get("SQ_WAIT_INST_ANY", 0) / life, 4)
            r["parked_share_of_wave_lifetime"] = round(m.get("SQ_WAIT_ANY", 0) / life, 4)
        if m.get("SQ_INSTS_VALU"):
            r["active_valu_quadcycles_per_valu_inst"] = round(m.get("SQ_ACTIVE_INST_VALU", 0) / m["SQ_INSTS_VALU"], 3)
            r["salu_per_valu"] = round(m.get("SQ_INSTS_SALU", 0) / m["SQ_INSTS_VALU"], 3)
        if k in dur:
            r["avg_us_trace_pass"] = round(dur[k] / 1e3, 2)
            r["clock_GHz"] = round(m.get("GRBM_GUI_ACTIVE", 0) / 8.0 / dur[k], 3)
            # vector instructions issued per SIMD per microsecond, all 1024 SIMDs
            r["valu_inst_per_simd_per_us"] = round(m.get("SQ_INSTS_VALU", 0) / 1024.0 / (dur[k] / 1e3), 1)
        out.append(r)
    return sorted(out, key=lambda r: -r.get("SQ_INSTS_VALU", 0))


res = {"note": __doc__, "bench_kernels": rows(load("pmc"), durations()), "calibration_kernels": rows(load("pmc_cal"), {})}
# the calibration binary prints its own timings: attach them in order of appearance
try:
    res["calibration_timings"] = open(os.path.join(src, "valu_rates.txt")).read().splitlines()
except OSError:
    pass
json.dump(res, open(os.path.join(ROOT, "profiles", tag + "_counters.json"), "w"), indent=1)
# what bench.py reads back as `valu_issue_frac` of the kernels whose inner loops have no closed instruction count (the
# event forms, the poly bank): vector instructions per launch, as counted here
json.dump({"source": "profiles/%s_counters.json (tools/prof_counters.sh %s: one rocprofv3 --pmc pass over bench.py)" % (tag, tag),
           "kernels": {r["kernel"]: {"SQ_INSTS_VALU": r.get("SQ_INSTS_VALU"), "SQ_INSTS_SALU": r.get("SQ_INSTS_SALU"),
                                      "launches": r["launches"], "avg_us_trace_pass": r.get("avg_us_trace_pass")}
                       for r in res["bench_kernels"]}},
          open(os.path.join(ROOT, "profiles", "counters.json"), "w"), indent=1)
for r in res["bench_kernels"]:
    print("%-60s x%-4d VALU %12.0f  share %.3f  stall %.3f  parked %.3f  clk %s  us %s" % (
        r["kernel"][:60], r["launches"], r.get("SQ_INSTS_VALU", 0), r.get("valu_active_share_of_wave_lifetime", 0),
        r.get("issue_stall_share_of_wave_lifetime", 0), r.get("parked_share_of_wave_lifetime", 0),
        r.get("clock_GHz"), r.get("avg_us_trace_pass")))
