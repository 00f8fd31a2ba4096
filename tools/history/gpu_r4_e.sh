#!/bin/bash
# round 4, call E (historical: the kernel was removed afterwards, commits 0b79420..23ea087 hold it): the software-pipelined MLP kernel (sherf_nerf_mlp_pipe) against the one-launch kernel: bit identity, timing, stress
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python tools/mlp_ab.py --config cfg2_dense_ri --rounds 3 --stress 100 --forms one,pipe --out $OUT/e_mlp_ab_dense.json > $OUT/e_mlp_ab_dense.log 2>&1; echo "[ab dense rc=$?]"; grep "^\[\|Error\|error" $OUT/e_mlp_ab_dense.log | cut -c1-220
timeout 300 python tools/mlp_ab.py --config cfg2_ri --rounds 3 --forms one,pipe --out $OUT/e_mlp_ab_ri.json > $OUT/e_mlp_ab_ri.log 2>&1; echo "[ab cfg2_ri rc=$?]"; grep "^\[\|Error\|error" $OUT/e_mlp_ab_ri.log | cut -c1-220
