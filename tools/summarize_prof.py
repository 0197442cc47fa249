"""tools/summarize_prof.py <tag> [key] -- copy the judged rocprofv3 summaries of
gpurun_out/prof_<tag>/ into profiles/ and fold the PMC traffic into profiles/traffic.json.

HBM bytes per launch follow MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are in
KiB, collected in separate --pmc passes; on gfx950 FETCH_SIZE counts wide (16 B/lane)
coalesced reads at exactly half, so reads = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact
for 16-B-per-lane stores."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

stats = max(glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")), key=os.path.getmtime)   # newest run
shutil.copy(stats, os.path.join(dst, tag + "_kernel_stats.csv"))
for leg in ("unprofiled", "trace", "fetch", "write"):
    p = os.path.join(src, "bench_%s.json" % leg)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, "%s_bench_%s.json" % (tag, leg)))


def pmc(leg, counter):
    f = max(glob.glob(os.path.join(src, "pmc_" + leg, "*", "*counter_collection.csv")), key=os.path.getmtime)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


fetch = pmc("fetch", "FETCH_SIZE")
write = pmc("write", "WRITE_SIZE")
bench = json.loads(open(os.path.join(src, "bench_trace.json")).read().strip().splitlines()[-1])
rows = []
for k in fetch:
    if "rocclr" in k:
        continue
    f_kib, n = fetch[k]
    w_kib, _ = write.get(k, (0.0, 0))
    rows.append({"kernel": k, "launches": n, "FETCH_SIZE_KiB_avg": f_kib, "WRITE_SIZE_KiB_avg": w_kib,
                 "hbm_read_bytes_per_launch": 2 * f_kib * 1024, "hbm_write_bytes_per_launch": w_kib * 1024,
                 "hbm_bytes_per_launch": 2 * f_kib * 1024 + w_kib * 1024})
summary = {"tag": tag, "bench_line_under_trace": bench, "pmc": rows,
           "note": "reads = 2*FETCH_SIZE*1024 (gfx950 half-count correction for 16 B/lane streams), "
                   "writes = WRITE_SIZE*1024; separate --pmc passes"}
json.dump(summary, open(os.path.join(dst, tag + "_pmc_traffic.json"), "w"), indent=1)

tpath = os.path.join(dst, "traffic.json")
tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
cfg = bench["config"]
key = sys.argv[2] if len(sys.argv) > 2 else "saw_v%d_f%d" % (cfg["voices_per_gpu"], cfg["frames_per_step"])
main = max(rows, key=lambda r: r["hbm_bytes_per_launch"] * r["launches"])   # the timed kernel, not one-off helpers
tj[key] = {"hbm_bytes_per_launch": main["hbm_bytes_per_launch"], "source": tag + "_pmc_traffic.json",
           "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"]}
json.dump(tj, open(tpath, "w"), indent=1)
print(json.dumps(rows, indent=1))
print(open(os.path.join(dst, tag + "_kernel_stats.csv")).read())
