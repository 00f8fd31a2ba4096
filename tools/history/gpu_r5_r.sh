#!/bin/bash
# round 5, call R: the backward's tall GEMMs, streaming kernel against the general one, shape by shape (bits + us), then the training step
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python tools/bwd_gemm_ab.py > $OUT/r5r_bwd_gemm_ab.txt 2>&1; echo "[rc=$?]"; cut -c1-260 $OUT/r5r_bwd_gemm_ab.txt | tail -14
timeout 600 python bench_train.py --steps 5 --warmup 2 > $OUT/r5r_train.json 2> $OUT/r5r_train.err; echo "[train rc=$?]"; python -c "
import json; d=json.loads(open('$OUT/r5r_train.json').read().strip().splitlines()[-1]); print({k: d.get(k) for k in ('ms_per_step','value','unit','stages_ms','host_ms')})"
SHERF_EXPERIMENT=64 timeout 600 python bench_train.py --steps 5 --warmup 2 > $OUT/r5r_train_general.json 2>> $OUT/r5r_train.err; echo "[train general rc=$?]"; python -c "
import json; d=json.loads(open('$OUT/r5r_train_general.json').read().strip().splitlines()[-1]); print({k: d.get(k) for k in ('ms_per_step','value','unit')})"
tail -3 $OUT/r5r_train.err | cut -c1-300
