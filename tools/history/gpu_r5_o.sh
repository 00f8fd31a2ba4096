#!/bin/bash
# round 5, call O: does the ~200 us idle stretch of the encoder's stream follow the STREAM (hardware queue) or the chain?  Side-stream pair swapped /
# default priority / two other queues, each in its own process (HIP-event timeline of the native driver), + a kernel timeline of the swapped pair.
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for X in 0 1 2 4 6; do
  SHERF_EXPERIMENT_STREAMS=$X timeout 300 python tools/frame_ab.py --config cfg2_dense_ri --arms 0 --names streams_$X --timeline --rounds 3 > $OUT/r5o_ab_$X.txt 2>&1; echo "[ab $X rc=$?]"
  grep "^\[arm\]\|^\[timeline\]" $OUT/r5o_ab_$X.txt | cut -c1-330
done
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc --no-train"
for X in 1 2; do
SHERF_EXPERIMENT_STREAMS=$X timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/r5o_prof -o trace -- $B > $OUT/r5o_prof.log 2>&1; echo "[rocprof rc=$?]"
DB=$(find $OUT/r5o_prof -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB > $OUT/r5o_timeline_streams_$X.txt 2>&1
rm -rf $OUT/r5o_prof
grep -n "scatter_rows\|step window\|mark_rows\|cand_search\|gather_tokens\|sconv3_kernel<1, 2, false, 0" $OUT/r5o_timeline_streams_$X.txt | cut -c1-150
done
