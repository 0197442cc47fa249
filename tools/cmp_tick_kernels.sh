export TMPDIR=/tmp
REPO=$PWD
for v in 33 26; do
  export SMX_SAW_TICK_MAX_LOG2=$v
  python tools/explore_tick2.py
  python bench.py --no-also --no-cpu --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  bench unprofiled: ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/cmp_$v -- python3 $REPO/bench.py --steps 200 --warmup 20 --no-also --no-cpu > /dev/null 2>&1)
  python - <<PY
import csv, glob
f=max(glob.glob('$REPO/gpurun_out/cmp_$v/*/*kernel_stats.csv'))
for r in csv.DictReader(open(f)):
    if 'saw_' in r['Name'] and int(r['Calls'])>100: print('  under rocprofv3 --kernel-trace:', r['Name'][28:60], r['Calls'], 'avg us', float(r['AverageNs'])/1e3, 'min', float(r['MinNs'])/1e3)
PY
done
