#!/bin/bash
# round 4, call S: frames issued round-robin on two caller streams (cross-frame overlap) with this round's shorter first phase
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
Q="--steps 40 --warmup 10 --no-cpu-baseline --no-torch-gpu-baseline --no-pmc --no-secondary --no-train"
{
for cfg in cfg2_dense_ri cfg2_ri; do
  for n in 1 2 3 1 2; do
    timeout 200 python bench.py --config $cfg --streams $n $Q > $OUT/s_bench.json 2> $OUT/s_bench.err; echo "[bench $cfg streams=$n rc=$?]"
    python -c "
import json; d=json.loads(open('$OUT/s_bench.json').read().strip().splitlines()[-1])
print('$cfg streams $n:', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,1), 'Mrays/s', d.get('parity_ok'), d['frame_timeline_ms'])"
  done
done
} > $OUT/r4_s.log 2>&1
cat $OUT/r4_s.log
