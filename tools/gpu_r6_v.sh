#!/bin/bash
# round 6, call V: the warp kernel as w persistent workgroups per CU (SHERF_EXPERIMENT bits 16-19: 3, 4, 5, 6) against one workgroup per 256 samples (eight resident per CU):
# does the encoder's 208-register convolution fit beside it and finish earlier?
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python tools/frame_ab.py --config cfg2_dense_ri --arms 0,0,0,0,0 --exps 0,0x30000,0x40000,0x50000,0x60000 --names base,w3,w4,w5,w6 --timeline --rounds 5 > $OUT/r6v_frame_ab.log 2>&1
echo "[ab rc=$?]"; grep "^\[timeline\|^\[arm\|^\[bits" $OUT/r6v_frame_ab.log | cut -c1-330
