#!/bin/bash
# round 5, call Q: the library GEMM as a yardstick for the network kernel's 0.44 (power cap): torch.matmul fp16 on normal data vs zeros
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 python tools/gemm_ceiling.py > $OUT/r5q_gemm_ceiling.txt 2>&1; echo "[rc=$?]"; cut -c1-250 $OUT/r5q_gemm_ceiling.txt
