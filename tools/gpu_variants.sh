#!/bin/bash
# kernel-variant comparison: alternative builds of the library (same ABI) selected with SHERF_HIP_LIB.
#   usage: bash tools/gpu_variants.sh [tag ...]      (tags of sherf_amd/libsherf_hip_<tag>.so; "" = the default build)
export SHERF_MLP_SHAPE=${SHERF_MLP_SHAPE:-8x1}   # A/B runs pin the MLP shape (bench.py would otherwise autotune it)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for V in "" "$@"; do
  export SHERF_HIP_LIB=$GRAFT_REPO_ROOT/sherf_amd/libsherf_hip${V:+_$V}.so
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | grep '"metric"\|rror' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant[$V]', 'ms/step %.3f' % d['ms_per_step'], 'mlp %.4f' % d['roofline']['kernel_ms'], 'frac %.4f' % d['roofline']['frac'])"
done
if [ -n "$1" ] && [ -n "$VARIANT_TESTS" ]; then
  export SHERF_HIP_LIB=$GRAFT_REPO_ROOT/sherf_amd/libsherf_hip_$1.so
  timeout 600 python -m pytest tests -m gpu -q --timeout=600 -x --no-header -p no:cacheprovider -s -k "per_sample or end_to_end" 2>&1 | grep -E "sigma\+|PSNR|passed|failed" | tail -12
fi
