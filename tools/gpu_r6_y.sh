#!/bin/bash
# round 6, call Y: does the ranking of the network's three launch forms IN the frame (behind the low-power first phase) match the ranking mlp_form='auto' measures with
# back-to-back launches?  The three forms pinned + auto, interleaved, timelines (mlp_ms) + the tuner's report of the same process
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python tools/frame_ab.py --config cfg2_dense_ri --arms 0,0,0,0 --names pipelined,two_tiles,one,auto --opts "mlp_form='pipelined';mlp_form='two_tiles';mlp_form='one';mlp_form='auto'" --timeline --rounds 6 > $OUT/r6y_frame_ab.log 2>&1
echo "[ab rc=$?]"; grep "^\[timeline\|^\[arm\|^\[bits\|^\[mlp_form" $OUT/r6y_frame_ab.log | cut -c1-330
