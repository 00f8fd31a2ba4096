#!/bin/bash
# round 6, call X: the two residency knobs of the first phase together -- search workgroups per CU (debug bits 20-23: 7, 8) x warp workgroups per CU (SHERF_EXPERIMENT bits 16-19: 4, 5, 6)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python tools/frame_ab.py --config cfg2_dense_ri --arms 0,0x700000,0x800000,0x800000,0x700000,0x800000 --exps 0,0x50000,0x40000,0x50000,0x60000,0x60000 --names base,s7w5,s8w4,s8w5,s7w6,s8w6 --timeline --rounds 5 > $OUT/r6x_frame_ab.log 2>&1
echo "[ab rc=$?]"; grep "^\[timeline\|^\[arm\|^\[bits" $OUT/r6x_frame_ab.log | cut -c1-330
