#!/bin/bash
# round 2, call G: backward GEMMs with the three-part bf16 split, LDS-resident sampler (8 / 12 / 16 waves, grid sizes), frame timeline,
# MLP start stagger (are the two workgroups of a CU phase-locked?)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python tools/mlp_trace.py > $OUT/mlp_trace_g.log 2>&1; echo "[trace rc=$?]"; grep "^\[lib\|^\[bf16\|^\[roof\|^\[trace\|^\[phase\|wave slots" $OUT/mlp_trace_g.log | cut -c1-220
timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -s -k "backward or gemm or elementwise or conv or mask_and_nearest or end_to_end or stage_by_stage or edge" > $OUT/pytest_g.log 2>&1; echo "[pytest rc=$?]"; tail -3 $OUT/pytest_g.log; grep "FAILED\|Error" $OUT/pytest_g.log | cut -c1-300 | head -20
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc"
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), d['frame_timeline_ms'])"; }
L=$GRAFT_REPO_ROOT/sherf_amd
$B 2>/dev/null | grep '"metric"' | pr nn8_default
SHERF_NN_GRID=192 $B 2>/dev/null | grep '"metric"' | pr nn8_grid192
SHERF_NN_GRID=128 $B 2>/dev/null | grep '"metric"' | pr nn8_grid128
SHERF_HIP_LIB=$L/libsherf_hip_nn16.so $B 2>/dev/null | grep '"metric"' | pr nn16_default
SHERF_HIP_LIB=$L/libsherf_hip_nn16.so SHERF_NN_GRID=192 $B 2>/dev/null | grep '"metric"' | pr nn16_grid192
SHERF_HIP_LIB=$L/libsherf_hip_nn12.so $B 2>/dev/null | grep '"metric"' | pr nn12_default
SHERF_HIP_LIB=$L/libsherf_hip_stag55.so $B 2>/dev/null | grep '"metric"' | pr stag55_frame

cd /tmp
prof() { # tag, env...
  local tag=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_g_$tag -o trace -- $B > $OUT/prof_g_$tag.log 2>&1; echo "[rocprof $tag rc=$?]"
  DB=$(find $OUT/prof_g_$tag -name "*.db" | head -1)
  if [ -n "$DB" ]; then
    python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 40 > $OUT/prof_g_${tag}_stats.txt
    python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB > $OUT/prof_g_${tag}_timeline.txt 2>&1
    grep "sample_nn\|gather_tokens\|nerf_mlp\|warp_geom" $OUT/prof_g_${tag}_stats.txt | cut -c1-110
    find $OUT/prof_g_$tag -name "*.db" -size +20M -delete
  fi
}
prof nn8 A=1
prof nn16 SHERF_HIP_LIB=$L/libsherf_hip_nn16.so
timeout 300 python $GRAFT_REPO_ROOT/bench_train.py --steps 3 --warmup 1 > $OUT/train_g.log 2>&1; echo "[train rc=$?]"; grep '"metric"' $OUT/train_g.log | cut -c1-600
