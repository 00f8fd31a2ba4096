#!/bin/bash
# round 5, call P: host clock at the frame driver's enqueue points while frames run back to back (is the host ahead of the GPU when the encoder's stream idles?),
# then timing events along the encoder's stream (us since the stream's entry into the frame)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 python tools/host_stamps.py > $OUT/r5p_host_stamps.txt 2>&1; echo "[rc=$?]"; cut -c1-300 $OUT/r5p_host_stamps.txt | head -14
for X in 32 33 34 40; do timeout 300 python tools/host_stamps.py --exp $X > $OUT/r5p_trail_$X.txt 2>&1; echo "[trail $X rc=$?]"; cut -c1-400 $OUT/r5p_trail_$X.txt | grep trail | tail -3; done
