#!/bin/bash
# round 5, call N: why does the encoder's stream sit idle for ~200 us behind level 0's scan?  Launch-order / stream-placement experiments
# (SHERF_EXPERIMENT, csrc/common.h) A/B'd in one process with the native driver's HIP-event timeline, then a kernel timeline of the best guess.
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 400 python tools/frame_ab.py --config cfg2_dense_ri --arms 0,0,0,0,0,0 --names base,scatter_first,levels_inline,both,encoder_first,enc_first+scatter --exps 0,1,2,3,8,9 --timeline --rounds 3 > $OUT/r5n_ab.txt 2>&1; echo "[ab rc=$?]"
grep "^\[bits\]\|^\[arm\]\|^\[timeline\]" $OUT/r5n_ab.txt | cut -c1-330
SHERF_EXPERIMENT_BASE=4 timeout 300 python tools/frame_ab.py --config cfg2_dense_ri --arms 0,0 --names dev_release,dev_release+scatter_first --exps 0,1 --timeline --rounds 3 > $OUT/r5n_ab_release.txt 2>&1; echo "[ab release rc=$?]"
grep "^\[bits\]\|^\[arm\]\|^\[timeline\]" $OUT/r5n_ab_release.txt | cut -c1-330
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc --no-train"
SHERF_EXPERIMENT=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/r5n_prof -o trace -- $B > $OUT/r5n_prof.log 2>&1; echo "[rocprof rc=$?]"
DB=$(find $OUT/r5n_prof -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB > $OUT/r5n_timeline_scatter_first.txt 2>&1
rm -rf $OUT/r5n_prof
grep -n "scatter_rows\|step window\|mark_rows\|cand_search\|gather_tokens" $OUT/r5n_timeline_scatter_first.txt | cut -c1-150
