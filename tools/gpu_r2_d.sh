#!/bin/bash
# round 2, call D: the MLP as two launches (tokens kernel at higher occupancy + decoder kernel), direct positional encodings
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
./tools/vsin_err
timeout 400 python tools/mlp_trace.py > $OUT/mlp_trace.log 2>&1; echo "[trace rc=$?]"; grep "^\[\|^ " $OUT/mlp_trace.log | head -40; tail -3 $OUT/mlp_trace.log | cut -c1-300
for L in "" slowmath; do
  SHERF_HIP_LIB=${L:+$GRAFT_REPO_ROOT/sherf_amd/libsherf_hip_$L.so} timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -s -k "margin_protocol" > $OUT/proto_$L.log 2>&1; echo "[protocol ${L:-product} rc=$?]"; grep "flips mask" $OUT/proto_$L.log | cut -c1-330
done
timeout 1200 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -s > $OUT/pytest_gpu.log 2>&1; echo "[pytest rc=$?]"; tail -5 $OUT/pytest_gpu.log; grep "flips mask\|FAILED" $OUT/pytest_gpu.log | cut -c1-330 | head -30
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "[bench rc=$?]"; cut -c1-2500 $OUT/bench_default.json; tail -3 $OUT/bench_default.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_d -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc > $OUT/prof_d.log 2>&1; echo "[rocprof rc=$?]"
cd $GRAFT_REPO_ROOT; DB=$(find $OUT/prof_d -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB 25 30 > $OUT/prof_d_stats.txt; head -24 $OUT/prof_d_stats.txt | cut -c1-150; find $OUT/prof_d -name "*.db" -size +20M -delete
