#!/bin/bash
# round 2, call A: every staged gpu_experimental test in its own process + bench_train.py (first hardware run of the backward)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
: > $OUT/exp.log
for T in $(python -m pytest tests/test_gpu_backward.py tests/test_gpu_tune.py -m gpu_experimental --collect-only -q -p no:cacheprovider 2>/dev/null | grep "::"); do
  timeout 240 python -m pytest "$T" -m gpu_experimental -q --no-header -p no:cacheprovider -x -s > $OUT/one.log 2>&1; rc=$?
  echo "[$rc] $T" | tee -a $OUT/exp.log
  if [ $rc -ne 0 ]; then tail -40 $OUT/one.log >> $OUT/exp.log; fi
done
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu_experimental -q -s --no-header -p no:cacheprovider > $OUT/ops.log 2>&1; echo "[ops rc=$?]" | tee -a $OUT/exp.log; tail -30 $OUT/ops.log >> $OUT/exp.log
timeout 300 python bench_train.py --steps 5 --warmup 2 > $OUT/bench_train.log 2>&1; echo "[bench_train rc=$?]" | tee -a $OUT/exp.log; tail -15 $OUT/bench_train.log >> $OUT/exp.log
