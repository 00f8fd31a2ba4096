#!/bin/bash
# round 4, call T: frames round-robin on 3 / 4 / 6 caller streams, default and 8 hardware queues
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
Q="--steps 48 --warmup 12 --no-cpu-baseline --no-torch-gpu-baseline --no-pmc --no-secondary --no-train"
{
for cfg in cfg2_dense_ri cfg2_ri; do
  for hq in 0 8; do
    for n in 1 3 4 6; do
      if [ $hq -gt 0 ]; then export GPU_MAX_HW_QUEUES=$hq; else unset GPU_MAX_HW_QUEUES; fi
      timeout 200 python bench.py --config $cfg --streams $n $Q > $OUT/t_bench.json 2> $OUT/t_bench.err; rc=$?
      python -c "
import json; d=json.loads(open('$OUT/t_bench.json').read().strip().splitlines()[-1])
print('$cfg hwq=$hq streams $n rc=$rc:', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,1), 'Mrays/s', 'frame_done', d['frame_timeline_ms']['frame_done'], 'mlp', round(d['roofline']['kernel_ms'],4))"
    done
  done
done
} > $OUT/r4_t.log 2>&1
cat $OUT/r4_t.log
