#!/bin/bash
# round 4, call K: gather + per-sample network in N parts on two streams (frame->mlp_parts), A/B against the whole-range schedule
mkdir -p gpurun_out
{
for cfg in cfg2_dense_ri cfg2_ri; do
  timeout 400 python tools/frame_ab.py --config $cfg --arms 0,0x20000,0x30000,0x40000,0x60000,0x80000 --names whole,parts2,parts3,parts4,parts6,parts8
done
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "full_size_frame_properties and dense" 2>&1 | tail -5
} > gpurun_out/r4_k.log 2>&1
tail -40 gpurun_out/r4_k.log
