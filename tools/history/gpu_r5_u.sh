#!/bin/bash
# round 5, call U: the new weight-gradient GEMM kernels against round 2's, shape by shape; tall GEMMs incl. the padded K = 71 / 199 operands; the training step
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python tools/bwd_gemm_ab.py > $OUT/r5u_bwd_gemm_ab.txt 2>&1; echo "[rc=$?]"; grep "^\[" $OUT/r5u_bwd_gemm_ab.txt | cut -c1-260; tail -3 $OUT/r5u_bwd_gemm_ab.txt | grep -v "^\[" | cut -c1-300
timeout 600 python bench_train.py --steps 5 --warmup 2 > $OUT/r5u_train.json 2> $OUT/r5u_train.err; echo "[train rc=$?]"; python -c "
import json; d=json.loads(open('$OUT/r5u_train.json').read().strip().splitlines()[-1]); print({k: d.get(k) for k in ('ms_per_step','value','unit','host_ms')})"
