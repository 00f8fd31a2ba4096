#!/bin/bash
# round 4, call P: shared corner look-ups in the gather: parity on the hardware, A/B against the previous library (libsherf_hip_prev.so)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
{
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -x -k "margin_protocol or per_sample or tokens" 2>&1 | tail -3
for cfg in cfg2_dense_ri cfg2_ri; do
  for lib in sherf_amd/libsherf_hip_prev.so sherf_amd/libsherf_hip.so sherf_amd/libsherf_hip_prev.so sherf_amd/libsherf_hip.so; do
    echo "== $cfg $lib"
    SHERF_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout 200 python tools/frame_ab.py --config $cfg --rounds 3 --timeline --arms 0 --names head 2>&1 | grep "timeline\|arm\|configuration"
  done
done
} > $OUT/r4_p.log 2>&1
cat $OUT/r4_p.log
