#!/bin/bash
# evidence refresh without the PMC passes: full GPU test-suite, smoke, default bench (with CPU baseline), rocprofv3 stats + timeline
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --timeout=600 --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 900 python bench.py > $OUT/bench_default.log 2>&1; echo "bench rc=$?"; grep '"metric"' $OUT/bench_default.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --bn-mode eval 2>/dev/null | grep '"metric"' > $OUT/bench_eval.log; cut -c1-260 $OUT/bench_eval.log
cd /tmp
rm -rf $OUT/prof; mkdir -p $OUT/prof
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/prof_bench.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $OUT/prof/t_results.db 25 60 > $OUT/kernel_stats.txt; head -14 $OUT/kernel_stats.txt | cut -c1-150
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $OUT/prof/t_results.db > $OUT/timeline.txt
grep '"metric"' $OUT/prof_bench.log | cut -c1-200
rm -rf $OUT/prof
