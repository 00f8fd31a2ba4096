#!/bin/bash
# round 5, call S: per-kernel totals of the training step at HEAD
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/r5s_prof -o trace -- python $GRAFT_REPO_ROOT/bench_train.py --steps 4 --warmup 1 > $OUT/r5s_prof.log 2>&1; echo "[rocprof rc=$?]"
DB=$(find $OUT/r5s_prof -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 70 > $OUT/r5s_train_stats.txt; rm -rf $OUT/r5s_prof; head -60 $OUT/r5s_train_stats.txt | cut -c1-150
