#!/bin/bash
# round 5, call E: is the network kernel power-bound?  The same launches on zeroed tokens (and zeroed weights): same instruction stream, less switching
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for z in "" tokens all; do
  timeout 300 python tools/mlp_ab.py --config cfg2_dense_ri --rounds 3 --forms one,tt,pp ${z:+--zero $z} --out $OUT/r5e_mlp_ab_zero_$z.json > $OUT/r5e_mlp_ab_zero_$z.log 2>&1; echo "[zero='$z' rc=$?]"; grep "^\[arm\|Error\|error" $OUT/r5e_mlp_ab_zero_$z.log | cut -c1-120
done
