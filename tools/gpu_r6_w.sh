#!/bin/bash
# round 6, call W: rocprofv3 --pmc passes (one counter set per pass, nothing else traced) of the bench frame: the sixteen-channel gather and the network kernel on the
# dense framing -- the same sets as round 5's call G (profiles/r05_call_g_pmc_network_and_gather_dense.txt) for a like-for-like reading
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
C="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --precision f16 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc --no-train"
echo "# round 6, call W: rocprofv3 --pmc passes of bench.py (cfg2_dense_ri, --precision f16, one frame in flight): per-kernel averages" > $OUT/r6w_pmc.txt
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  TAG=$(echo $SET | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $SET -d $OUT/r6w_pmc_$TAG -o pmc -- $C > $OUT/r6w_pmc_$TAG.log 2>&1; echo "[pmc $TAG rc=$?]"
  DB=$(find $OUT/r6w_pmc_$TAG -name "*.db" | head -1)
  echo "[pmc $TAG]" >> $OUT/r6w_pmc.txt
  [ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/pmc_query.py $DB gather_tokens nerf_mlp 2>&1 | cut -c1-120 | grep -v "^# pmc" >> $OUT/r6w_pmc.txt
  rm -rf $OUT/r6w_pmc_$TAG
done
cat $OUT/r6w_pmc.txt | head -90
