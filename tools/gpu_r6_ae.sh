#!/bin/bash
# round 6, call AE: four frames in flight after the fix (frames on several caller streams are read back every frame again: the host throttle), and one frame in flight
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="--steps 100 --no-secondary --no-train --no-cpu-baseline --no-torch-gpu-baseline --no-pmc"
for rep in 1 2; do
  for n in 4 1; do
    timeout 300 python bench.py --streams $n $B > $OUT/r6ae_$n.json 2> $OUT/r6ae_$n.err; python -c "
import json; d=json.loads(open('$OUT/r6ae_$n.json').read().strip().splitlines()[-1]); print('streams $n', d['ms_per_step'], d['config'].get('mlp_form'))"
  done
done
