#!/bin/bash
# round 6, call G: (1) the capture-pattern probe on torch's bundled HIP runtime (the one the product process runs on); (2) frames as graphs WITHOUT the
# third stream (the three-stream capture crashes hipStreamEndCapture in-process, the two-stream one works) against eager frames with / without it
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TL=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
ls $TL | grep -i "amdhip\|hsa-runtime" | head
for v in 2 3 4 7; do LD_LIBRARY_PATH=$TL timeout 60 tools/ubench/graph_probe $v 2>&1 | tail -4; echo "[torch-runtime variant $v rc=$?]"; done | tee $OUT/r6g_graph_probe_torch_runtime.txt
timeout 900 python tools/frame_ab.py --config cfg2_dense_ri --arms 0,0,0 --names eager_aux,eager_noaux,graph_noaux --opts "frame_graph=False;frame_graph=False,aux_stream=False;frame_graph=True,aux_stream=False" --timeline --rounds 4 > $OUT/r6g_frame_ab.log 2>&1
echo "[frame_ab rc=$?]"; grep "^\[\|configuration" $OUT/r6g_frame_ab.log | cut -c1-400
