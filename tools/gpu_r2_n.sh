#!/bin/bash
# round 2, call N: sparse conv with compile-time modes + pinned refills (new) against the previous form (sconvold), sampler occupancy cap
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc"
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), d['frame_timeline_ms'])"; }
L=$GRAFT_REPO_ROOT/sherf_amd
for pad in 0 10240 18432 26624; do
  SHERF_NN_LDS_PAD=$pad $B 2>/dev/null | grep '"metric"' | pr new_pad$pad
done
for pad in 0 10240; do
  SHERF_HIP_LIB=$L/libsherf_hip_sconvold.so SHERF_NN_LDS_PAD=$pad $B 2>/dev/null | grep '"metric"' | pr old_pad$pad
done
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -k "sparse_voxel or end_to_end or mask_and_nearest" > $OUT/pytest_n.log 2>&1; echo "[pytest rc=$?]"; tail -2 $OUT/pytest_n.log
SHERF_HIP_LIB=$L/libsherf_hip_sconvtrace.so timeout 300 python tools/sconv_trace.py > $OUT/sconv_trace_n.log 2>&1; echo "[sconv trace rc=$?]"; grep -A18 "encoder alone" $OUT/sconv_trace_n.log | awk '{print $1,$2,$3,$4,$5,$7,$8,$12,$13,$16}' | head -20
