#!/bin/bash
# round 6, call AH: the two per-frame memsets in whole 16-byte blocks (one fill kernel each instead of two): frame_ab cannot switch it -- kernel trace of the bench frame
# (fills per frame, timeline) and the bench line
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc --no-train"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_ah -o trace -- $B > $OUT/prof_ah.log 2>&1; echo "[rocprof rc=$?]"
DB=$(find $OUT/prof_ah -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 60 20 > $OUT/r6ah_prof_stats.txt; grep -n "fillBuffer" $OUT/r6ah_prof_stats.txt | cut -c1-150
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB > $OUT/r6ah_timeline.txt 2>&1; grep -c "fillBuffer" $OUT/r6ah_timeline.txt; grep "fillBuffer\|build_cells2\|near_lists_pairs" $OUT/r6ah_timeline.txt | head -12 | cut -c1-120
rm -rf $OUT/prof_ah
cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 300 python bench.py --no-secondary --no-train --no-cpu-baseline --no-torch-gpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['frame_timeline_ms'])"; done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "stage or mask or voxel or determin" 2>&1 | tail -2
