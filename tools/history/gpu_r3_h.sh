#!/bin/bash
# round 3, call H: the training step with the binned tap scatter
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_backward.py -m gpu -q --no-header -p no:cacheprovider > $OUT/h_pytest.log 2>&1; echo "[pytest rc=$?]"
tail -3 $OUT/h_pytest.log | cut -c1-300; grep "^FAILED\|^ERROR\|^E  " $OUT/h_pytest.log | cut -c1-300 | head -12
timeout 300 python bench_train.py --steps 4 --warmup 2 > $OUT/h_train.json 2> $OUT/h_train.err; echo "[train rc=$?]"; cut -c1-1500 $OUT/h_train.json; tail -3 $OUT/h_train.err | cut -c1-300
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/h_prof -o trace -- python $GRAFT_REPO_ROOT/bench_train.py --steps 3 --warmup 1 > $OUT/h_prof.log 2>&1; echo "[rocprof rc=$?]"
DB=$(find $OUT/h_prof -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 45 > $OUT/h_prof_stats.txt; head -36 $OUT/h_prof_stats.txt | cut -c1-170
find $OUT/h_prof -name "*.db" -size +20M -delete
