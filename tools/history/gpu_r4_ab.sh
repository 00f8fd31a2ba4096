#!/bin/bash
# round 4, call AB: the multi-rank path of bench.py with four caller streams per rank, dry-run on ONE GPU: 2 ranks on device 0, gloo instead of
# RCCL (two RCCL ranks cannot share a device) -- the launch, the per-stream frames, the asynchronous gathers awaited a step later, the
# barrier / max-over-ranks timing and the JSON line; both partitions
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export SHERF_LOCAL_DEVICE=0 SHERF_DIST_BACKEND=gloo SHERF_DIST_TIMEOUT=120
{
for part in views rays; do
  timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 12 --warmup 4 --partition $part --config cfg2_ri > $OUT/ab_bench.json 2> $OUT/ab_bench.err; rc=$?
  echo "[partition $part rc=$rc]"
  python -c "
import json
ls=[l for l in open('$OUT/ab_bench.json').read().splitlines() if l.startswith('{')]
d=json.loads(ls[-1]); print({k: d.get(k) for k in ('value','ms_per_step','n_gpus','rccl_ranks','scaling')}, d['config'].get('caller_streams'), d['config'].get('parallelism'), d.get('roofline', {}).get('kernel_ms'))" || tail -5 $OUT/ab_bench.err | cut -c1-400
done
} > $OUT/r4_ab.log 2>&1
cat $OUT/r4_ab.log
