#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python tools/gpu_diag.py tiny > $OUT/diag.log 2>&1; echo "diag rc=$?"
grep -v amdgpu.ids $OUT/diag.log | tail -25
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -x --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 $OUT/pytest.log
rm -rf $OUT/prof; mkdir -p $OUT/prof
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/prof_bench.log 2>&1
echo "prof rc=$?"; grep '"metric"' $OUT/prof_bench.log
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $OUT/prof/trace_results.db 13 45 > $OUT/kernel_stats.txt; rm -rf $OUT/prof
cut -c1-150 $OUT/kernel_stats.txt | head -32
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep '"metric"' | tee $OUT/bench.log
