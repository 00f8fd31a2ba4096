#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python tools/gpu_diag.py tiny > $OUT/diag.log 2>&1; echo "diag rc=$?"
grep -E "nv hip|mismatch|bf16x3|psnr|EXCEPTION|Error" $OUT/diag.log | tail -8
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -x --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 $OUT/pytest.log
cd /tmp
rm -rf $OUT/prof; mkdir -p $OUT/prof
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/prof_bench.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $OUT/prof/t_results.db 13 45 > $OUT/kernel_stats.txt; cut -c1-150 $OUT/kernel_stats.txt | head -22; grep '"metric"' $OUT/prof_bench.log | cut -c1-200
rm -rf $OUT/prof
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf $OUT/pmc; mkdir -p $OUT/pmc
  timeout 300 rocprofv3 --kernel-trace --pmc $P -d $OUT/pmc -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc.log 2>&1; echo "pmc rc=$? ($P)"
  python $GRAFT_REPO_ROOT/tools/pmc_query.py $OUT/pmc/p_results.db nerf_mlp gather_tokens sample_nn >> $OUT/pmc_summary.txt 2>&1
  rm -rf $OUT/pmc
done
cat $OUT/pmc_summary.txt | cut -c1-160 | head -90
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep '"metric"' | tee $OUT/bench.log | cut -c1-260
