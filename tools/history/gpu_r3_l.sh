#!/bin/bash
# round 3, call L: what the empty workgroups of the capacity-sized MLP launch cost (MLP alone at the exact grid; the frame with --exact-grids)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
Q="--steps 40 --warmup 10 --no-cpu-baseline --no-torch-gpu-baseline --no-pmc"
run() { timeout 300 python bench.py $Q $1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d.get('frame_timeline_ms'); s=(d.get('secondary') or {}).get('mlp_kernel_alone') or {}
print('$1', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,1), 'Mrays/s mlp in frame', round(d['roofline']['kernel_ms'],4), 'alone', {k: round(v['kernel_ms'],4) for k,v in s.items() if isinstance(v, dict) and 'kernel_ms' in v}, t)"; }
run "--precision f16"
run "--precision f16 --exact-grids --no-secondary"
run "--precision f16 --no-secondary"
