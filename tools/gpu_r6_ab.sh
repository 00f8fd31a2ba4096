#!/bin/bash
# round 6, call AB: the compositing kernel walking a ray's samples in batches of four (loads of a batch requested together; same arithmetic in the same order) against
# the one-sample-per-trip loop of rounds 1-5 (SHERF_EXPERIMENT bit 20), interleaved; kernel trace of both
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python tools/frame_ab.py --config cfg2_dense_ri --arms 0,0,0,0 --exps 0x100000,0,0x100000,0 --names one,batch4,one2,batch4b --timeline --rounds 6 > $OUT/r6ab_frame_ab.log 2>&1
echo "[ab rc=$?]"; grep "^\[timeline\|^\[arm\|^\[bits" $OUT/r6ab_frame_ab.log | cut -c1-330
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc --no-train"
for xp in 0 1048576; do
  SHERF_EXPERIMENT=$xp timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_ab$xp -o trace -- $B > $OUT/prof_ab$xp.log 2>&1; echo "[rocprof $xp rc=$?]"
  DB=$(find $OUT/prof_ab$xp -name "*.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 45 20 | grep "composite_compact" | cut -c1-150
  rm -rf $OUT/prof_ab$xp
done
