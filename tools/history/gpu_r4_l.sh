#!/bin/bash
# round 4, call L: exact near lists per sub-cell for the candidate search (sherf_build_near_lists): bit-exact ids / order on the hardware,
# A/B against the cell walk (sherf_set_debug bit 14) in both framings, kernel trace of the dense frame
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
{
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -x -k "mask_and_nearest or warp_matches or ragged or deterministic or margin_protocol or no_valid" 2>&1 | tail -4
for cfg in cfg2_dense_ri cfg2_ri; do
  timeout 300 python tools/frame_ab.py --config $cfg --arms 0,0x4000 --names lists,cellwalk --rounds 3
done
} > $OUT/r4_l.log 2>&1
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc --no-train"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/l_prof -o trace -- $B > $OUT/l_prof.log 2>&1; echo "[rocprof rc=$?]" >> $OUT/r4_l.log
DB=$(find $OUT/l_prof -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 45 > $OUT/l_prof_stats.txt; python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB > $OUT/l_prof_timeline.txt 2>&1
head -16 $OUT/l_prof_stats.txt | cut -c1-150 >> $OUT/r4_l.log
grep -i "near_lists\|cand_\|near_mask\|build_cells" $OUT/l_prof_stats.txt | cut -c1-150 >> $OUT/r4_l.log
tail -3 $OUT/l_prof.log | cut -c1-1500 >> $OUT/r4_l.log
find $OUT/l_prof -name "*.db" -size +20M -delete
cat $OUT/r4_l.log
