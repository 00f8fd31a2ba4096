#!/bin/bash
# round 4, call Z: the fp16-table gather compiled for 5 / 6 waves per SIMD (96 / 80 VGPRs) against the shipped 4 (110 VGPRs)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
Q="--steps 48 --warmup 12 --no-cpu-baseline --no-torch-gpu-baseline --no-pmc --no-secondary --no-train"
{
for lib in libsherf_hip.so libsherf_hip_gu.so libsherf_hip_gu4.so libsherf_hip.so libsherf_hip_gu.so libsherf_hip_gu4.so; do
  for n in 1 4; do
    SHERF_HIP_LIB=$GRAFT_REPO_ROOT/sherf_amd/$lib timeout 200 python bench.py --config cfg2_dense_ri --streams $n $Q > $OUT/z_bench.json 2> $OUT/z_bench.err; rc=$?
    python -c "
import json; d=json.loads(open('$OUT/z_bench.json').read().strip().splitlines()[-1])
t=d.get('frame_timeline_ms') or {}
print('$lib streams $n rc=$rc:', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,1), 'Mrays/s', 'gather', round(t.get('gather_done',0)-t.get('rays_at_encoder_join',0),4) if $n==1 else '', d.get('parity_ok'))"
  done
done
} > $OUT/r4_z.log 2>&1
cat $OUT/r4_z.log
