#!/bin/bash
# round 4, call AA: frames in flight x streams per frame (with / without the third, "aux", stream)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
Q="--steps 48 --warmup 12 --no-cpu-baseline --no-torch-gpu-baseline --no-pmc --no-secondary --no-train --config cfg2_dense_ri"
{
for aux in 1 0; do
  for n in 4 5 6 4; do
    SHERF_AUX_STREAM=$aux timeout 200 python bench.py --streams $n $Q > $OUT/aa_bench.json 2> $OUT/aa_bench.err; rc=$?
    python -c "
import json; d=json.loads(open('$OUT/aa_bench.json').read().strip().splitlines()[-1])
print('aux=$aux streams $n rc=$rc:', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,1), 'Mrays/s')"
  done
done
} > $OUT/r4_aa.log 2>&1
cat $OUT/r4_aa.log
