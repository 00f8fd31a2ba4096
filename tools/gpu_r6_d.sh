#!/bin/bash
# round 6, call D: where does the capture of a frame crash?  (call C: segmentation fault inside sherf_render_frame at the second sighting.)  The enqueue
# points are traced on stderr (SHERF_FRAME_GRAPH_DEBUG); then the tiny backward test's encoder figures with the scaled-lo input-gradient convolutions
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export SHERF_FRAME_GRAPH_DEBUG=1
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu --no-header -p no:cacheprovider -s -k "hipgraphs" > $OUT/r6d_pytest_graphs.log 2>&1
echo "[pytest graphs rc=$?]"; grep "sherf\]" $OUT/r6d_pytest_graphs.log | tail -25; tail -3 $OUT/r6d_pytest_graphs.log | cut -c1-300
unset SHERF_FRAME_GRAPH_DEBUG
SHERF_FRAME_GRAPH=0 timeout 900 python -m pytest tests/test_gpu_backward.py -q -m gpu --no-header -p no:cacheprovider -s \
  -k "full_backward_against_reference or (full_size_backward and dense)" > $OUT/r6d_pytest_backward.log 2>&1
echo "[pytest backward rc=$?]"; grep "encoder gradients vs\|norm-relative\|encoder_3d\|worst outside\|vertex_feat\|passed\|failed\|Error" $OUT/r6d_pytest_backward.log | cut -c1-200 | head -40
