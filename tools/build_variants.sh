#!/bin/bash
# Builds alternative libraries (same ABI) with kernel-variant macros / flags for A/B runs (tools/mlp_trace.py, SHERF_HIP_LIB=...):
#   bash tools/build_variants.sh [tag ...]   -> sherf_amd/libsherf_hip_<tag>.so   (git-ignored, travel with gpurun)
set -e
cd "$(dirname "$0")/.."
python -m sherf_amd.build >/dev/null
cd sherf_amd
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-value -mcode-object-version=5"
build() { # tag, source file (without .hip), defines...
  local tag=$1 src=$2; shift; shift
  local extra=""; { [ "$src" = mlp ] || [ "$src" = gather ]; } && extra="-fno-slp-vectorize"      # (sherf_amd/build.py: EXTRA_FLAGS)
  /opt/rocm/bin/hipcc $FLAGS $extra "$@" -c csrc/$src.hip -o build/variant_$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v "/$src.o\|variant_\|bwd_\|ops_") build/variant_$tag.o -o libsherf_hip_$tag.so
  echo "built libsherf_hip_$tag.so ($src: $*)"
}
declare -A DEFS=( [trace]="-DSHERF_MLP_TRACE=1" [prio0]="-DSHERF_MLP_DECODER_PRIO=0" [slowmath]="-DSHERF_MLP_FASTMATH=0 -DSHERF_MLP_FAST_ERF=0" [nomix]="-DSHERF_MLP_FMA_MIX=0"
                  [nopkrelu]="-DSHERF_MLP_PK_RELU=0" [lb4]="-DSHERF_MLP_LB=4" [nodma]="-DSHERF_MLP_ABLATE=32" [nobar]="-DSHERF_MLP_ABLATE=64" [sconvtrace]="-DSHERF_SCONV_TRACE=1" [tw3]="-DSHERF_MLP_TOKENS_WAVES=3" [tw5]="-DSHERF_MLP_TOKENS_WAVES=5" [noperm]="-DSHERF_MLP_PERMLANE=0" [nodmabar]="-DSHERF_MLP_ABLATE=96" [noall]="-DSHERF_MLP_ABLATE=224" [notrans]="-DSHERF_MLP_ABLATE=256" [nope]="-DSHERF_MLP_ABLATE=512" [nocvt]="-DSHERF_MLP_ABLATE=1024" [novalu]="-DSHERF_MLP_ABLATE=1792" [ceiling]="-DSHERF_MLP_ABLATE=2016" [prio0]="-DSHERF_MLP_DECODER_PRIO=0" [mfmaonly]="-DSHERF_MLP_ABLATE=2816" [noepi]="-DSHERF_MLP_ABLATE=2048" [gnoslp]="-fno-slp-vectorize" [snoslp]="-fno-slp-vectorize" [vnoslp]="-fno-slp-vectorize" [nnrows]="-DSHERF_NN_ROWS_VARIANT=2" )
declare -A SRC=( [sconvtrace]=svox [gnoslp]=gather [snoslp]=sample [vnoslp]=svox [nnrows]=sample )
TAGS=${@:-trace nodma}
for t in $TAGS; do build $t ${SRC[$t]:-mlp} ${DEFS[$t]} & done
wait
