#!/bin/bash
# round 5, call D: the decoder with its epilogues inside the MFMA stream (sherf_nerf_mlp3, form `pp`) against the one-tile and two-tile kernels
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 400 python tools/mlp_ab.py --config cfg2_dense_ri --rounds 3 --stress 60 --forms one,tt,pp --out $OUT/r5d_mlp_ab_dense.json > $OUT/r5d_mlp_ab_dense.log 2>&1; echo "[ab dense rc=$?]"; grep "^\[\|Error\|error" $OUT/r5d_mlp_ab_dense.log | cut -c1-220
timeout 300 python tools/mlp_ab.py --config cfg2_ri --rounds 3 --forms one,tt,pp --out $OUT/r5d_mlp_ab_ri.json > $OUT/r5d_mlp_ab_ri.log 2>&1; echo "[ab cfg2_ri rc=$?]"; grep "^\[\|Error\|error" $OUT/r5d_mlp_ab_ri.log | cut -c1-220
