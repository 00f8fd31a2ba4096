#!/bin/bash
# round 6, call K: gather.hip without SLP vectorisation (now the product build) + 32-bit table offsets, and the voxel rows of the next corner requested ahead
# (SHERF_EXPERIMENT bit 9; the first run of this call used debug bit 21, which also sets the candidate search's workgroups per CU): whole-frame A/B in one process, bits + timeline
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python tools/frame_ab.py --config cfg2_dense_ri --arms 0,0 --exps 0,512 --names base,prefetch --timeline --rounds 4 > $OUT/r6k_frame_ab.log 2>&1
echo "[frame_ab rc=$?]"; grep "^\[\|configuration" $OUT/r6k_frame_ab.log | cut -c1-400
timeout 900 python tools/frame_ab.py --config cfg2_ri --arms 0,0 --exps 0,512 --names base,prefetch --rounds 3 > $OUT/r6k_frame_ab_cfg2.log 2>&1
echo "[frame_ab cfg2_ri rc=$?]"; grep "^\[arm\|^\[bits" $OUT/r6k_frame_ab_cfg2.log | cut -c1-400
