#!/bin/bash
# round 5, call C: in-kernel timeline of the two-tile network kernel + SQ counters of both kernels on the dense frame
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python tools/mlp2_trace.py --out $OUT/r5c_mlp2_trace.json > $OUT/r5c_mlp2_trace.log 2>&1; echo "[trace rc=$?]"; grep "^\[\|^   \|Error\|error" $OUT/r5c_mlp2_trace.log | cut -c1-600
