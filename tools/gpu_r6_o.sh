#!/bin/bash
# round 6, call O: the candidate search's workgroups per CU re-swept (debug bits 20-23; 6 = the default since round 4) now that the second phase is shorter
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python tools/frame_ab.py --config cfg2_dense_ri --arms 0,0x400000,0x500000,0x700000,0x800000 --names wg6,wg4,wg5,wg7,wg8 --timeline --rounds 3 > $OUT/r6o_frame_ab.log 2>&1
echo "[frame_ab rc=$?]"; grep "^\[\|configuration" $OUT/r6o_frame_ab.log | cut -c1-330
