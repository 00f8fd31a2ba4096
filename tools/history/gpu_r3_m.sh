#!/bin/bash
# round 3, call M: tri-plane / pixel taps gathered before the encoder join (gather_split) now that the ray side reaches the join first
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
Q="--steps 40 --warmup 10 --no-cpu-baseline --no-torch-gpu-baseline --no-pmc --no-secondary --precision f16"
run() { env $1 timeout 300 python bench.py $Q $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d.get('frame_timeline_ms')
print('$1 $2', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,1), 'Mrays/s parity_ok', d.get('parity_ok'), t)"; }
run X=0 ""
run SHERF_GATHER_SPLIT=1 ""
run X=0 ""
run SHERF_GATHER_SPLIT=1 ""
run SHERF_GATHER_SPLIT=1 "--config cfg2_dense_ri"
run X=0 "--config cfg2_dense_ri"
