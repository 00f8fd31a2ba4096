#!/bin/bash
# round 6, call C: frame graphs captured on the library's own stream (call B: every capture failed -- the caller's stream was the legacy default
# stream), and the input-gradient convolutions with both operands' lo halves at 2^11 against the float64 truth at full size
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export SHERF_FRAME_GRAPH_DEBUG=1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_producers.py -q -m gpu --no-header -p no:cacheprovider -s \
  -k "hipgraphs or graphed_producer" > $OUT/r6c_pytest_graphs.log 2>&1
echo "[pytest graphs rc=$?]"; grep -i "sherf\] frame graph" $OUT/r6c_pytest_graphs.log | sort | uniq -c | head -5; tail -4 $OUT/r6c_pytest_graphs.log | cut -c1-300
timeout 600 python tools/frame_ab.py --config cfg2_dense_ri --arms 0,0 --names eager,graph --opts "frame_graph=False;frame_graph=True" --timeline --rounds 4 > $OUT/r6c_frame_ab.log 2>&1
echo "[frame_ab rc=$?]"; grep "^\[\|configuration" $OUT/r6c_frame_ab.log | sort | uniq -c | cut -c1-400 | head -20
timeout 600 python tools/frame_ab.py --config cfg2_ri --arms 0,0 --names eager,graph --opts "frame_graph=False;frame_graph=True" --rounds 3 > $OUT/r6c_frame_ab_cfg2.log 2>&1
echo "[frame_ab cfg2_ri rc=$?]"; grep "^\[arm\|^\[graphs\|^\[bits" $OUT/r6c_frame_ab_cfg2.log | cut -c1-400
timeout 1500 python -m pytest tests/test_gpu_backward.py -q -m gpu --no-header -p no:cacheprovider -s \
  -k "mfma_input_gradient or full_backward_against_reference or (full_size_backward and cfg2_ri)" > $OUT/r6c_pytest_backward.log 2>&1
echo "[pytest backward rc=$?]"; grep "input-gradient conv\|norm-relative\|encoder_3d\|worst outside\|vertex_feat\|passed\|failed" $OUT/r6c_pytest_backward.log | cut -c1-200 | head -40
