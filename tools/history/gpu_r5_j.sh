#!/bin/bash
# round 5, call J: gather + network in parts on two streams again, now with the pipelined network at REDUCED residency (two / one workgroups per CU: registers and
# LDS left for the next part's gather beside it) -- the network is power-bound, its own duration should not depend on the residency
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python tools/frame_ab.py --config cfg2_dense_ri --arms 0,0,0,0,0,0x4000000,0x8000000,0x8000000 --names whole,p2,p3,p4,p6,p4res3,p4res1,p8res1 \
   --opts ";mlp_parts=2;mlp_parts=3;mlp_parts=4;mlp_parts=6;mlp_parts=4;mlp_parts=4;mlp_parts=8" --timeline --rounds 3 > $OUT/r5j_frame_ab_parts.txt 2>&1; echo "[frame_ab rc=$?]"
grep "^\[arm\|^\[bits\|^\[timeline\|configuration\|Error" $OUT/r5j_frame_ab_parts.txt | cut -c1-260
