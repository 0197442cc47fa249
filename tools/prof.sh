#!/bin/bash
# tools/prof.sh <tag> -- rocprofv3 kernel trace + PMC traffic passes of the bench command.
# Run on the GPU box from the repo root; writes gpurun_out/prof_<tag>/.
set -e
TAG=${1:-r01}
shift || true
ARGS="${@:---steps 50 --warmup 5 --no-also --no-cpu}"
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
cd $REPO
find $OUT -name "*.csv" | head -20
