#!/bin/bash
# round 6, call AF: frames the host may run ahead with four caller streams (SHERF_WATCH_RING = slots of the read-back ring: 2, 3, 4, 6, 8) and caller streams (3, 4, 6)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="--steps 100 --no-secondary --no-train --no-cpu-baseline --no-torch-gpu-baseline --no-pmc"
for cfg in "4 2" "4 3" "4 4" "4 6" "4 8" "3 3" "6 6" "2 2" "4 4"; do
  set -- $cfg
  SHERF_WATCH_RING=$2 timeout 300 python bench.py --streams $1 $B > $OUT/r6af.json 2> $OUT/r6af.err; python -c "
import json; d=json.loads(open('$OUT/r6af.json').read().strip().splitlines()[-1]); print('streams $1 ring $2', round(d['ms_per_step'], 4))"
done
