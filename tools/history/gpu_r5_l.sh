#!/bin/bash
# round 5, call L: the gather with the two coarser levels' occupancy records staged in LDS (debug bit 29; bit 30: 16 instead of 4 pairs of tiles per workgroup)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python tools/frame_ab.py --config cfg2_dense_ri --arms 0,0x20000000,0x60000000 --names default,stage4,stage16 --timeline --rounds 4 > $OUT/r5l_frame_ab_gather_stage.txt 2>&1; echo "[frame_ab dense rc=$?]"
grep "^\[arm\|^\[bits\|^\[timeline\|Error" $OUT/r5l_frame_ab_gather_stage.txt | cut -c1-260
timeout 400 python tools/frame_ab.py --config cfg2_ri --arms 0,0x20000000,0x60000000 --names default,stage4,stage16 --rounds 3 > $OUT/r5l_frame_ab_gather_stage_ri.txt 2>&1; echo "[frame_ab cfg2_ri rc=$?]"
grep "^\[arm\|^\[bits\|Error" $OUT/r5l_frame_ab_gather_stage_ri.txt | cut -c1-260
