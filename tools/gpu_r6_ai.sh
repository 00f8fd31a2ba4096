#!/bin/bash
# round 6, call AI: the frame's depth range (first read by the compositing) computed on the stream that carries the SMPL tables instead of in the ray side's chain
# (sherf_depth_range; 2 launches off the critical path): parity / glue / ray-tile tests, the bench line twice, the timeline of a profiled frame
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 300 python bench.py --no-secondary --no-train --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('parity_ok'), d['frame_timeline_ms'])"; done
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_glue.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc --no-train"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_ai -o trace -- $B > $OUT/prof_ai.log 2>&1; echo "[rocprof rc=$?]"
DB=$(find $OUT/prof_ai -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB > $OUT/r6ai_timeline.txt 2>&1; grep "depth_minmax\|init_range\|init_counters\|cand_mark\|near_lists_pairs_kernel<true>" $OUT/r6ai_timeline.txt | head -8 | cut -c1-120
rm -rf $OUT/prof_ai
