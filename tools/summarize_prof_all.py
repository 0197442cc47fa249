"""tools/summarize_prof_all.py <tag> -- fold gpurun_out/prof_<tag>_all/ (tools/prof_all.sh) into
profiles/<tag>_all_kernel_stats.csv, profiles/<tag>_all_kernels_traffic.json, profiles/<tag>_bench_full.json.
HBM bytes: reads = 2 * FETCH_SIZE KiB (gfx950 half-count correction for 16 B/lane streams), writes =
WRITE_SIZE KiB (MI355X_MICROARCH.md, HBM / rocprofv3 section); separate --pmc passes."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "prof_%s_all" % tag)
dst = os.path.join(ROOT, "profiles")
stats = max(glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")), key=os.path.getmtime)   # newest run
shutil.copy(stats, os.path.join(dst, tag + "_all_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench_trace.json"), os.path.join(dst, tag + "_bench_full_under_trace.json"))
dur = {r["Name"]: (float(r["AverageNs"]), int(r["Calls"])) for r in csv.DictReader(open(stats))}


def pmc(leg, counter):
    f = max(glob.glob(os.path.join(src, "pmc_" + leg, "*", "*counter_collection.csv")), key=os.path.getmtime)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
rows = []
for k, (ns, calls) in sorted(dur.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    if "rocclr" in k:
        continue
    rd, wr = 2 * fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
    rows.append({"kernel": k, "launches": calls, "avg_us": round(ns / 1e3, 2), "hbm_read_MB": round(rd / 1e6, 2),
                 "hbm_write_MB": round(wr / 1e6, 2), "GBps_at_trace_duration": round((rd + wr) / ns, 1)})
json.dump({"note": "whole bench.py run (headline + also-workloads); reads = 2*FETCH_SIZE KiB (gfx950 correction, "
                   "valid for 16 B/lane streams; narrower accesses are uncalibrated), writes = WRITE_SIZE KiB; "
                   "per-kernel averages over all launches of that kernel in the run (a kernel used at several "
                   "sizes is an average over them); durations from the separate kernel-trace pass",
           "kernels": rows}, open(os.path.join(dst, tag + "_all_kernels_traffic.json"), "w"), indent=1)
for r in rows:
    print("%-70s x%-4d %9.1f us  R %9.1f MB  W %9.1f MB  %7.1f GB/s" % (r["kernel"][:70], r["launches"], r["avg_us"],
          r["hbm_read_MB"], r["hbm_write_MB"], r["GBps_at_trace_duration"]))
