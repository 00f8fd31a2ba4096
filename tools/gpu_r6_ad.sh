#!/bin/bash
# round 6, call AD: four frames in flight regressed in evidence run 4 (1.535 ms against 1.37-1.38 in runs 1-3): which change?  The four-stream bench line with the
# watch reading every frame / every eighth, with the form pinned (no tuner), twice each
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="python bench.py --streams 4 --steps 100 --no-secondary --no-train --no-cpu-baseline --no-torch-gpu-baseline --no-pmc"
for rep in 1 2; do
for tag in "head" "watch1:SHERF_WATCH_EVERY=1" "pinned:SHERF_MLP_FORM=two_tiles" "both:SHERF_WATCH_EVERY=1 SHERF_MLP_FORM=two_tiles"; do
  name=${tag%%:*}; envs=""; [ "$tag" != "$name" ] && envs=${tag#*:}
  env $envs timeout 300 $B > $OUT/r6ad_$name.json 2> $OUT/r6ad_$name.err; python -c "
import json; d=json.loads(open('$OUT/r6ad_$name.json').read().strip().splitlines()[-1]); print('$name', d['ms_per_step'], d['config'].get('mlp_form'), d['config'].get('caller_streams'))"
done
done
