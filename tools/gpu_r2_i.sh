#!/bin/bash
# round 2, call I: sparse-conv prefetch pipeline, 14-dword sampler records, batched warp NN, gather fused into the MLP kernel (A/B)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_glue.py tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -s -k "exact_grids_and_gather or mask_and_nearest or end_to_end or stage_by_stage or warp or sparse_voxel or gathered" > $OUT/pytest_i.log 2>&1; echo "[pytest rc=$?]"; tail -3 $OUT/pytest_i.log; grep "FAILED\|Error" $OUT/pytest_i.log | cut -c1-300 | head -20
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc"
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), d['frame_timeline_ms'])"; }
$B 2>/dev/null | grep '"metric"' | pr fused
SHERF_SPLIT_GATHER=1 $B 2>/dev/null | grep '"metric"' | pr split
$B --exact-grids 2>/dev/null | grep '"metric"' | pr fused_exact_grids
SHERF_SPLIT_GATHER=1 $B --exact-grids 2>/dev/null | grep '"metric"' | pr split_exact_grids
cd /tmp
prof() { # tag, env...
  local tag=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_i_$tag -o trace -- $B > $OUT/prof_i_$tag.log 2>&1; echo "[rocprof $tag rc=$?]"
  DB=$(find $OUT/prof_i_$tag -name "*.db" | head -1)
  if [ -n "$DB" ]; then
    python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 40 > $OUT/prof_i_${tag}_stats.txt
    python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB > $OUT/prof_i_${tag}_timeline.txt 2>&1
    grep "sample_nn\|gather_tokens\|nerf_mlp\|warp_geom\|sconv3\|compact_k" $OUT/prof_i_${tag}_stats.txt | cut -c1-130
    find $OUT/prof_i_$tag -name "*.db" -size +20M -delete
  fi
}
prof fused A=1
prof split SHERF_SPLIT_GATHER=1
cd $GRAFT_REPO_ROOT
SHERF_HIP_LIB=$GRAFT_REPO_ROOT/sherf_amd/libsherf_hip_sconvtrace.so timeout 300 python tools/sconv_trace.py > $OUT/sconv_trace_i.log 2>&1; echo "[sconv trace rc=$?]"; grep -v "^/opt" $OUT/sconv_trace_i.log | cut -c1-230 | head -24
