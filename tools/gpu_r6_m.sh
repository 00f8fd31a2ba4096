#!/bin/bash
# round 6, call M: the compaction with one lane per ray (default) against one wave per ray (SHERF_EXPERIMENT bit 11): whole-frame A/B, bits + timeline; kernel trace of a few frames
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python tools/frame_ab.py --config cfg2_dense_ri --arms 0,0 --exps 0,2048 --names lane_per_ray,wave_per_ray --timeline --rounds 4 > $OUT/r6m_frame_ab.log 2>&1
echo "[frame_ab rc=$?]"; grep "^\[\|configuration" $OUT/r6m_frame_ab.log | cut -c1-400
timeout 900 python tools/frame_ab.py --config cfg2_ri --arms 0,0 --exps 0,2048 --names lane_per_ray,wave_per_ray --rounds 3 > $OUT/r6m_frame_ab_cfg2.log 2>&1
echo "[frame_ab cfg2_ri rc=$?]"; grep "^\[arm\|^\[bits" $OUT/r6m_frame_ab_cfg2.log | cut -c1-400
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc --no-train"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_m -o trace -- $B > $OUT/prof_m.log 2>&1; echo "[rocprof rc=$?]"
DB=$(find $OUT/prof_m -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 24 > $OUT/r6m_prof_stats.txt; head -28 $OUT/r6m_prof_stats.txt | cut -c1-150
rm -rf $OUT/prof_m
