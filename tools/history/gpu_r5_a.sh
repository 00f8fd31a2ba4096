#!/bin/bash
# round 5, call A: the two-tiles-per-wave network kernel (sherf_nerf_mlp2, form `tt`) against the one-tile kernel: bit identity, timing, stress
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 400 python tools/mlp_ab.py --config cfg2_dense_ri --rounds 3 --stress 60 --forms one,tt --out $OUT/r5a_mlp_ab_dense.json > $OUT/r5a_mlp_ab_dense.log 2>&1; echo "[ab dense rc=$?]"; grep "^\[\|Error\|error" $OUT/r5a_mlp_ab_dense.log | cut -c1-220
timeout 300 python tools/mlp_ab.py --config cfg2_ri --rounds 3 --forms one,tt --out $OUT/r5a_mlp_ab_ri.json > $OUT/r5a_mlp_ab_ri.log 2>&1; echo "[ab cfg2_ri rc=$?]"; grep "^\[\|Error\|error" $OUT/r5a_mlp_ab_ri.log | cut -c1-220
