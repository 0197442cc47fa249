#!/bin/bash
# tools/prof.sh <tag> -- rocprofv3 kernel trace + PMC traffic passes of the bench command.
# Run on the GPU box from the repo root; writes gpurun_out/prof_<tag>/.
set -e
TAG=${1:-r01}
shift || true
ARGS="${@:---steps 200 --warmup 20 --no-also --no-cpu}"   # the default bench command (its step counts), headline only
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
# a fresh box runs its first GPU process ~5 % slow (measured: the same kernel 79.6 us in the first process, 75.3 us in
# the second, profiled or not): one un-profiled run of the same command first, kept beside the profiled ones
python3 $REPO/bench.py $ARGS > $OUT/bench_unprofiled.json 2> $OUT/unprofiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
cd $REPO
find $OUT -name "*.csv" | head -20
