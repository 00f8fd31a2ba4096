#!/bin/bash
# round 6, call S: gather + network cut into 2 / 3 / 4 parts on two streams (debug bits 16-19; the parts take the eight-channel gather) against the one-part frame with the
# sixteen-channel gather, interleaved in one process
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python tools/frame_ab.py --config cfg2_dense_ri --arms 0,0x20000,0x30000,0x40000 --names whole,2parts,3parts,4parts --timeline --rounds 5 > $OUT/r6s_frame_ab.log 2>&1
echo "[ab rc=$?]"; grep "^\[timeline\|^\[arm\|^\[bits" $OUT/r6s_frame_ab.log | cut -c1-330
