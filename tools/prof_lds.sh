#!/bin/bash
# tools/prof_lds.sh <tag> [bench args] -- rocprofv3 PMC passes (counters only, no trace with them) that split LDS time
# from vector time in the kernels of one bench leg (default: config 4, the poly bank): SQ_INSTS_LDS,
# SQ_ACTIVE_INST_LDS, SQ_LDS_BANK_CONFLICT next to SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES.
# Run on the GPU box from the repo root; the CSVs land in gpurun_out/prof_<tag>_lds/.
set -e
TAG=${1:-r03}
shift || true
ARGS="${@:---no-cpu --no-verify --steps 5 --warmup 2 --repeats 1 --legs c4}"
OUT=$PWD/gpurun_out/prof_${TAG}_lds
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i "lds" > $OUT/lds_counters_available.txt || true
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc1 -- python3 $REPO/bench.py $ARGS > $OUT/bench1.json 2> $OUT/bench1.err || echo "pass 1 failed" >&2
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc2 -- python3 $REPO/bench.py $ARGS > $OUT/bench2.json 2> $OUT/bench2.err || echo "pass 2 failed" >&2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
cd $REPO
find $OUT -name "*counter_collection.csv" | head
