#!/bin/bash
# round 3, call J: the training step with the register-summed binned scatter and the fused epilogues
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
echo skip-pytest
tail -3 $OUT/j_pytest.log | cut -c1-300; grep "^FAILED\|^ERROR\|^E  " $OUT/j_pytest.log | cut -c1-300 | head -12
timeout 300 python bench_train.py --steps 4 --warmup 2 > $OUT/j_train.json 2> $OUT/j_train.err; echo "[train rc=$?]"; cut -c1-1500 $OUT/j_train.json; tail -3 $OUT/j_train.err | cut -c1-300
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/j_prof -o trace -- python $GRAFT_REPO_ROOT/bench_train.py --steps 3 --warmup 1 > $OUT/j_prof.log 2>&1; echo "[rocprof rc=$?]"
DB=$(find $OUT/j_prof -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 45 > $OUT/j_prof_stats.txt; head -36 $OUT/j_prof_stats.txt | cut -c1-170
find $OUT/j_prof -name "*.db" -size +20M -delete
