#!/bin/bash
# round 4, call W: with four frames in flight the SUM of the kernels counts: search workgroups per CU again (8 / 6 / 5 / 7)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
Q="--steps 48 --warmup 12 --no-cpu-baseline --no-torch-gpu-baseline --no-pmc --no-secondary --no-train"
{
for cfg in cfg2_dense_ri cfg2_ri; do
  for dbg in 8388608 7340032 5242880 0 8388608; do
    SHERF_DEBUG=$dbg timeout 200 python bench.py --config $cfg --streams 4 $Q > $OUT/w_bench.json 2> $OUT/w_bench.err; rc=$?
    python -c "
import json; d=json.loads(open('$OUT/w_bench.json').read().strip().splitlines()[-1])
print('$cfg debug=$dbg rc=$rc:', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,1), 'Mrays/s')"
  done
done
} > $OUT/r4_w.log 2>&1
cat $OUT/r4_w.log
