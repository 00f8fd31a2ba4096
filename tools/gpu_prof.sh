#!/bin/bash
# kernel-trace profile of the bench command (no counters in this pass)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT/prof
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/prof_bench.log 2>&1
echo "prof rc=$?"; tail -3 $OUT/prof_bench.log
find $OUT/prof -name "*kernel_stats*" | head
F=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'total kernel time {tot/1e6:.2f} ms over all steps')
for r in rows[:45]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.3f} ms  {r['Calls']:>6} calls  avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f}%  {r['Name'][:110]}")
PY
