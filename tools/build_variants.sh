#!/bin/bash
# Builds alternative libraries (same ABI) with kernel-variant macros for A/B runs with tools/gpu_variants.sh:
#   bash tools/build_variants.sh            -> sherf_amd/libsherf_hip_{il8,erf,erf_il8}.so   (git-ignored, travel with gpurun)
#   gpurun -- 'bash tools/gpu_variants.sh il8 erf erf_il8'      (VARIANT_TESTS=1 also runs the per-sample parity tests on $1)
set -e
cd "$(dirname "$0")/../sherf_amd"
python -m sherf_amd.build >/dev/null 2>&1 || (cd .. && python -m sherf_amd.build >/dev/null)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-value -mcode-object-version=5"
build() { # tag, defines...
  local tag=$1; shift
  /opt/rocm/bin/hipcc $FLAGS "$@" -c csrc/mlp.hip -o build/mlpv_$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v "mlp\|bwd_") build/mlpv_$tag.o -o libsherf_hip_$tag.so
  echo "built libsherf_hip_$tag.so ($*)"
}
build il8 -DSHERF_MLP_INTERLEAVE=8
build erf -DSHERF_MLP_FAST_ERF=1
build erf_il8 -DSHERF_MLP_FAST_ERF=1 -DSHERF_MLP_INTERLEAVE=8
