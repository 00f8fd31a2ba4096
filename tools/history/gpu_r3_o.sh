#!/bin/bash
# round 3, call O: the backward's GPU tests and the training step after the element-wise kernels' rework
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_backward.py -m gpu -q --no-header -p no:cacheprovider > $OUT/o_pytest.log 2>&1; echo "[pytest rc=$?]"; tail -3 $OUT/o_pytest.log | cut -c1-300; grep "^FAILED\|^ERROR\|^E  " $OUT/o_pytest.log | cut -c1-300 | head
timeout 300 python bench_train.py --steps 4 --warmup 2 > $OUT/o_train.json 2> $OUT/o_train.err; echo "[train rc=$?]"; cut -c1-700 $OUT/o_train.json
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/o_prof -o trace -- python $GRAFT_REPO_ROOT/bench_train.py --steps 3 --warmup 1 > $OUT/o_prof.log 2>&1; echo "[rocprof rc=$?]"
DB=$(find $OUT/o_prof -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 45 > $OUT/o_prof_stats.txt; head -24 $OUT/o_prof_stats.txt | cut -c1-150
find $OUT/o_prof -name "*.db" -size +20M -delete
