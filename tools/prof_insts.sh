#!/bin/bash
# tools/prof_insts.sh <tag> <bench args...> -- one rocprofv3 PMC pass with instruction counters
# (SQ_INSTS_VALU, SQ_INSTS_SALU, SQ_WAVES) to check the ops-per-voice-sample figures of DESIGN.md.
set -e
TAG=$1; shift
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $OUT/pmc_insts -- python3 $REPO/bench.py "$@" --no-also --no-cpu > $OUT/bench_insts.json 2> $OUT/insts.err
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
f = glob.glob(out + "/pmc_insts/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
d = json.loads(open(out + "/bench_insts.json").read().strip().splitlines()[-1])
vs = d["config"]["voices_per_gpu"] * d["config"]["frames_per_step"]
res = {"voice_samples_per_launch": vs, "kernels": {}}
for k, c in agg.items():
    if "rocclr" in k: continue
    res["kernels"][k[:90]] = {n: sum(v) / len(v) for n, v in c.items()}
    res["kernels"][k[:90]]["launches"] = len(next(iter(c.values())))
    for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU"):
        if n in c:
            # counters count wave-instructions; one wave-instruction = 64 lane-ops
            res["kernels"][k[:90]][n + "_per_voice_sample"] = sum(c[n]) / len(c[n]) * 64 / vs
json.dump(res, open(out + "/insts_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
