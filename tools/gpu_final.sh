#!/bin/bash
# round-end style evidence run: tests, smoke, default bench (with CPU baseline), kernel-trace stats + timeline, PMC of the hot kernels
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --timeout=600 --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 900 python bench.py > $OUT/bench_default.log 2>&1; echo "bench rc=$?"; grep '"metric"' $OUT/bench_default.log | cut -c1-2500
timeout 300 python bench.py --no-cpu-baseline --bn-mode eval 2>/dev/null | grep '"metric"' > $OUT/bench_eval.log; cut -c1-260 $OUT/bench_eval.log
cd /tmp
rm -rf $OUT/prof; mkdir -p $OUT/prof
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/prof_bench.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $OUT/prof/t_results.db 25 60 > $OUT/kernel_stats.txt; head -14 $OUT/kernel_stats.txt | cut -c1-150
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $OUT/prof/t_results.db > $OUT/timeline.txt
grep '"metric"' $OUT/prof_bench.log | cut -c1-200
rm -rf $OUT/prof
rm -f $OUT/pmc_summary.txt
for P in "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"; do
  rm -rf $OUT/pmc; mkdir -p $OUT/pmc
  timeout 300 rocprofv3 --kernel-trace --pmc $P -d $OUT/pmc -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc.log 2>&1; echo "pmc rc=$? ($P)"
  python $GRAFT_REPO_ROOT/tools/pmc_query.py $OUT/pmc/p_results.db nerf_mlp gather_tokens sample_nn sconv3 >> $OUT/pmc_summary.txt 2>&1
  rm -rf $OUT/pmc
done
grep -A12 "nerf_mlp" $OUT/pmc_summary.txt | cut -c1-120 | head -44
