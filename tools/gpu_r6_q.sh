#!/bin/bash
# round 6, call Q: the two 96-column single-product sparse convolutions as three 32-column workgroups per row tile (SHERF_EXPERIMENT bit 12; svox.hip,
# sconv3_kernel: CS) against the whole-row instances: interleaved frames in one process (bit-for-bit compare by frame_ab), timelines, then the kernel trace
# of the bench frame with the bit set
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python tools/frame_ab.py --config cfg2_dense_ri --arms 0,0,0,0 --exps 0,4096,0,4096 --names whole,split,whole2,split2 --timeline --rounds 6 > $OUT/r6q_frame_ab.log 2>&1
echo "[ab rc=$?]"; grep "^\[timeline\|^\[arm\|identical\|differ" $OUT/r6q_frame_ab.log | cut -c1-330
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc --no-train"
for xp in 0 4096; do
  SHERF_EXPERIMENT=$xp timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_q$xp -o trace -- $B > $OUT/prof_q$xp.log 2>&1; echo "[rocprof $xp rc=$?]"
  DB=$(find $OUT/prof_q$xp -name "*.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 40 > $OUT/r6q_prof_stats_$xp.txt; grep -i "sconv3\|nerf_mlp\|gather_tokens" $OUT/r6q_prof_stats_$xp.txt | cut -c1-170
  rm -rf $OUT/prof_q$xp
done
