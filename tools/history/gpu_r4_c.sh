#!/bin/bash
# round 4, call C: VALU ablations of the MLP kernel (no transformer arithmetic / no positional encodings / cheap epilogues / all three /
# all three + no DMA, barriers, LDS re-reads = the pure MFMA stream), priority off; SQ counters of the product kernel on the dense frame.
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python tools/mlp_ab.py --config cfg2_dense_ri --rounds 3 --out $OUT/c_mlp_ab_dense.json > $OUT/c_mlp_ab_dense.log 2>&1; echo "[ab dense rc=$?]"; grep "^\[" $OUT/c_mlp_ab_dense.log | cut -c1-220
cd /tmp
C="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --precision f16 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc"
for form in 0 1; do
SHERF_MLP_SPLIT=$form timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $OUT/c_pmc1_$form -o pmc -- $C > $OUT/c_pmc1_$form.log 2>&1; echo "[pmc1 split=$form rc=$?]"
DB=$(find $OUT/c_pmc1_$form -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/pmc_query.py $DB nerf_ > $OUT/c_pmc1_$form.txt 2>&1; cat $OUT/c_pmc1_$form.txt | cut -c1-120 | head -40; find $OUT/c_pmc1_$form -name "*.db" -size +20M -delete
SHERF_MLP_SPLIT=$form timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM -d $OUT/c_pmc2_$form -o pmc -- $C > $OUT/c_pmc2_$form.log 2>&1; echo "[pmc2 split=$form rc=$?]"
DB=$(find $OUT/c_pmc2_$form -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/pmc_query.py $DB nerf_ > $OUT/c_pmc2_$form.txt 2>&1; cat $OUT/c_pmc2_$form.txt | cut -c1-120 | head -40; find $OUT/c_pmc2_$form -name "*.db" -size +20M -delete
done
