#!/bin/bash
# round 6, call AG: one frame in flight: does bounding how far the host runs ahead (ring slots x frames between read-backs) matter there too?
cd $GRAFT_REPO_ROOT
B="--steps 200 --no-secondary --no-train --no-cpu-baseline --no-torch-gpu-baseline --no-pmc"
for cfg in "4 8" "1 8" "2 8" "1 4" "1 2" "4 8"; do
  set -- $cfg
  SHERF_WATCH_RING=$1 SHERF_WATCH_EVERY=$2 timeout 300 python bench.py --streams 1 $B 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ring $1 every $2', round(d['ms_per_step'], 4), d['frame_timeline_ms']['host_wall_per_step_in_timed_loop'])"
done
