#!/bin/bash
# round 6, call U: the kernel trace of the bench command with the table over the timed frames only (tools/rocpd_stats.py tail_n) beside the live HIP-event figure
# of the same process (the bench's JSON line)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc --no-train"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_u -o trace -- $B > $OUT/prof_u.log 2>&1; echo "[rocprof rc=$?]"
DB=$(find $OUT/prof_u -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 45 20 > $OUT/r6u_prof_stats.txt
grep -n "nerf_mlp\|gather_tokens_h16\|timed frames" $OUT/r6u_prof_stats.txt | cut -c1-150
grep "^{" $OUT/prof_u.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench under the profiler:', d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
grep "^{" $OUT/prof_u.log | tail -1 > $OUT/r6u_bench_under_profiler.json
rm -rf $OUT/prof_u
