#!/bin/bash
# tools/prof_all.sh <tag> -- rocprofv3 over the WHOLE bench.py run (headline + also-workloads):
# one kernel-trace pass and two PMC passes (FETCH_SIZE, WRITE_SIZE; never combined with a trace).
# Run on the GPU box from the repo root; writes gpurun_out/prof_<tag>_all/; fold into profiles/ with
#   python3 tools/summarize_prof_all.py <tag>
set -e
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_${TAG}_all
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --no-cpu > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace pass done" >&2
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --no-cpu > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo "fetch pass done" >&2
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --no-cpu > $OUT/bench_write.json 2> $OUT/write.err
echo "write pass done" >&2
cd $REPO
find $OUT -name "*.csv" < /dev/null | head -20
