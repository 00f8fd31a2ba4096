#!/bin/bash
# round 4, call Y: with four frames in flight -- the ray side started behind encoder layers (less contention inside a frame, the overlap comes from other frames)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
Q="--steps 48 --warmup 12 --no-cpu-baseline --no-torch-gpu-baseline --no-pmc --no-secondary --no-train --streams 4"
{
for cfg in cfg2_dense_ri; do
  for mal in -1 0 3 7 12 -1; do
    SHERF_MAIN_AFTER_LAYER=$mal timeout 200 python bench.py --config $cfg $Q > $OUT/y_bench.json 2> $OUT/y_bench.err; rc=$?
    python -c "
import json; d=json.loads(open('$OUT/y_bench.json').read().strip().splitlines()[-1])
print('$cfg main_after_layer=$mal rc=$rc:', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,1), 'Mrays/s')"
  done
  for gs in 1; do
    SHERF_GATHER_SPLIT=$gs timeout 200 python bench.py --config $cfg $Q > $OUT/y_bench.json 2> $OUT/y_bench.err; rc=$?
    python -c "
import json; d=json.loads(open('$OUT/y_bench.json').read().strip().splitlines()[-1])
print('$cfg gather_split=$gs rc=$rc:', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,1), 'Mrays/s')"
  done
done
} > $OUT/r4_y.log 2>&1
cat $OUT/r4_y.log
