#!/bin/bash
# tools/prof_counters.sh <tag> [bench args] -- ONE rocprofv3 PMC pass (no trace with it) over bench.py with the
# instruction / busy counters of the SQ block and the GRBM clock counter, for every kernel of the run, plus the same
# pass over tools/ubench/valu_rates (kernels that are issue-bound by construction: the calibration of "busy").
# Run on the GPU box from the repo root; fold into profiles/ with  python3 tools/summarize_counters.py <tag>
set -e
TAG=${1:-r02}
shift || true
ARGS="${@:---no-cpu --no-verify --steps 30 --warmup 5}"
OUT=$PWD/gpurun_out/prof_${TAG}_counters
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
CTRS="SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
cd /tmp
rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc -- python3 $REPO/bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
echo "bench pass done" >&2
rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc_cal -- $REPO/tools/ubench/valu_rates > $OUT/valu_rates.txt 2> $OUT/cal.err
echo "calibration pass done" >&2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace pass done" >&2
cd $REPO
find $OUT -name "*.csv" < /dev/null | head
