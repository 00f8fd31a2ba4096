#!/bin/bash
# round 6, call H: frames as graphs WITHOUT the third stream against eager frames with / without it (bits, timeline, host time)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python tools/frame_ab.py --config cfg2_dense_ri --arms 0,0,0 --names eager_aux,eager_noaux,graph_noaux --opts "frame_graph=False;frame_graph=False,aux_stream=False;frame_graph=True,aux_stream=False" --timeline --rounds 4 > $OUT/r6h_frame_ab.log 2>&1
echo "[frame_ab rc=$?]"; grep "^\[\|configuration" $OUT/r6h_frame_ab.log | cut -c1-400
timeout 900 python tools/frame_ab.py --config cfg2_ri --arms 0,0,0 --names eager_aux,eager_noaux,graph_noaux --opts "frame_graph=False;frame_graph=False,aux_stream=False;frame_graph=True,aux_stream=False" --rounds 3 > $OUT/r6h_frame_ab_cfg2.log 2>&1
echo "[frame_ab cfg2_ri rc=$?]"; grep "^\[arm\|^\[graphs\|^\[bits" $OUT/r6h_frame_ab_cfg2.log | cut -c1-400
SHERF_FRAME_GRAPH=0 timeout 900 python -m pytest tests/test_gpu_backward.py -q -m gpu --no-header -p no:cacheprovider -s \
  -k "mfma_input_gradient or full_backward_against_reference or full_size_backward" > $OUT/r6h_pytest_backward.log 2>&1
echo "[pytest backward rc=$?]"; grep "encoder gradients vs\|norm-relative\|encoder_3d\|worst outside\|vertex_feat\|passed\|failed\|Error" $OUT/r6h_pytest_backward.log | cut -c1-200 | head -40
