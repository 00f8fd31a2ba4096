#!/bin/bash
# Builds alternative libraries (same ABI) with kernel-variant macros for A/B runs with tools/gpu_variants.sh:
#   bash tools/build_variants.sh      -> sherf_amd/libsherf_hip_{il8,erf,erf_il8,gbl,prio,prio_il8}.so   (git-ignored, travel with gpurun)
#   gpurun -- 'bash tools/gpu_variants.sh il8 erf erf_il8 gbl'   (VARIANT_TESTS=1 also runs the per-sample parity tests on $1)
set -e
cd "$(dirname "$0")/.."
python -m sherf_amd.build >/dev/null
cd sherf_amd
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-value -mcode-object-version=5"
build() { # tag, source file (without .hip), defines...
  local tag=$1 src=$2; shift; shift
  /opt/rocm/bin/hipcc $FLAGS "$@" -c csrc/$src.hip -o build/variant_$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v "/$src.o\|variant_\|bwd_") build/variant_$tag.o -o libsherf_hip_$tag.so
  echo "built libsherf_hip_$tag.so ($src: $*)"
}
build il8 mlp -DSHERF_MLP_INTERLEAVE=8
build erf mlp -DSHERF_MLP_FAST_ERF=1
build erf_il8 mlp -DSHERF_MLP_FAST_ERF=1 -DSHERF_MLP_INTERLEAVE=8
build gbl gather -DSHERF_GATHER_BRANCHLESS=1
build prio mlp -DSHERF_MLP_WAVE_PRIO=2
build prio_il8 mlp -DSHERF_MLP_WAVE_PRIO=2 -DSHERF_MLP_INTERLEAVE=8
