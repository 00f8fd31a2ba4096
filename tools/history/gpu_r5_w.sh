#!/bin/bash
# round 5, call W: training step + kernel totals (quick loop while the backward is being worked on)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python bench_train.py --steps 5 --warmup 2 > $OUT/r5w_train.json 2> $OUT/r5w_train.err; echo "[train rc=$?]"; python -c "
import json; d=json.loads(open('$OUT/r5w_train.json').read().strip().splitlines()[-1]); print({k: d.get(k) for k in ('ms_per_step','value','unit','host_ms')})"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/r5w_prof -o trace -- python $GRAFT_REPO_ROOT/bench_train.py --steps 4 --warmup 1 > $OUT/r5w_prof.log 2>&1; echo "[rocprof rc=$?]"
DB=$(find $OUT/r5w_prof -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 70 > $OUT/r5w_train_stats.txt; rm -rf $OUT/r5w_prof; head -${1:-16} $OUT/r5w_train_stats.txt | cut -c1-150
