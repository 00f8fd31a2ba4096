#!/bin/bash
# round 6, call T: the encoder's running-statistics update launched on the encoder's stream (beside the gather) instead of behind the compositing on the caller's
# stream (rendering option bn_update_on_side, default True), interleaved with the old placement in one process; then the GPU tests that read the running statistics
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python tools/frame_ab.py --config cfg2_dense_ri --arms 0,0,0,0 --names tail,side,tail2,side2 --opts "bn_update_on_side=False;bn_update_on_side=True;bn_update_on_side=False;bn_update_on_side=True" --timeline --rounds 6 > $OUT/r6t_frame_ab.log 2>&1
echo "[ab rc=$?]"; grep "^\[timeline\|^\[arm\|^\[bits" $OUT/r6t_frame_ab.log | cut -c1-330; tail -3 $OUT/r6t_frame_ab.log | cut -c1-300
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_glue.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
