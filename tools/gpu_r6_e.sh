#!/bin/bash
# round 6, call E: which stream-capture pattern crashes hipStreamEndCapture on this runtime?  (tools/ubench/graph_probe.hip, one process per variant)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for v in 1 2 3 4 5 6 7; do timeout 60 tools/ubench/graph_probe $v 2>&1 | tail -8; echo "[variant $v rc=$?]"; done | tee $OUT/r6e_graph_probe.txt
