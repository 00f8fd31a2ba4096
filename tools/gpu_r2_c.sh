#!/bin/bash
# round 2, call C: the rewritten MLP kernel (v2: two accumulator chains, 3-slot ring, f16x3) on hardware: variants + in-kernel
# timeline, the whole -m gpu suite, the default bench line, a kernel-trace profile
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 400 python tools/mlp_trace.py > $OUT/mlp_trace.log 2>&1; echo "[trace rc=$?]"; grep "^\[\|^ " $OUT/mlp_trace.log | head -40; tail -3 $OUT/mlp_trace.log
timeout 1200 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x -s > $OUT/pytest_gpu.log 2>&1; echo "[pytest rc=$?]"; tail -5 $OUT/pytest_gpu.log; grep "flips mask\|rel-to-max\|GB/s\|PSNR" $OUT/pytest_gpu.log | cut -c1-330 | head -30
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "[bench rc=$?]"; cut -c1-3000 $OUT/bench_default.json; tail -3 $OUT/bench_default.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_c -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc > $OUT/prof_c.log 2>&1; echo "[rocprof rc=$?]"
cd $GRAFT_REPO_ROOT; find $OUT/prof_c -type f | head; DB=$(find $OUT/prof_c -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB 25 30 > $OUT/prof_c_stats.txt; head -32 $OUT/prof_c_stats.txt; find $OUT/prof_c -name "*.db" -size +20M -delete
