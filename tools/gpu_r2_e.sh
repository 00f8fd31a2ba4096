#!/bin/bash
# round 2, call E: the re-fused MLP kernel (direct encodings, leaner prologue); sampler ablation; training-step profile; new tests
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 400 python tools/mlp_trace.py > $OUT/mlp_trace.log 2>&1; echo "[trace rc=$?]"; grep "^\[\|^ " $OUT/mlp_trace.log | head -30 | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -s > $OUT/pytest_gpu.log 2>&1; echo "[pytest rc=$?]"; tail -6 $OUT/pytest_gpu.log; grep "flips mask\|conditioning\|FAILED\|Error" $OUT/pytest_gpu.log | cut -c1-330 | head -40
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_e -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc > $OUT/prof_e.log 2>&1; echo "[rocprof rc=$?]"
SHERF_DEBUG=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_e_nocand -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc > $OUT/prof_e_nocand.log 2>&1; echo "[rocprof nocand rc=$?]"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_e_train -o trace -- python $GRAFT_REPO_ROOT/bench_train.py --steps 3 --warmup 1 > $OUT/prof_e_train.log 2>&1; echo "[rocprof train rc=$?]"; tail -2 $OUT/prof_e_train.log | cut -c1-600
cd $GRAFT_REPO_ROOT
for d in prof_e prof_e_nocand prof_e_train; do DB=$(find $OUT/$d -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB 0 40 > $OUT/${d}_stats.txt; echo "== $d"; head -22 $OUT/${d}_stats.txt | cut -c1-160; find $OUT/$d -name "*.db" -size +20M -delete; done
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "[bench rc=$?]"; cut -c1-1800 $OUT/bench_default.json; tail -3 $OUT/bench_default.err
