#!/bin/bash
# round 4, call N: slimmer host side + six search workgroups per CU: A/B, quick bench lines of both framings (timeline, host per step)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
{
for cfg in cfg2_dense_ri cfg2_ri; do
  timeout 300 python tools/frame_ab.py --config $cfg --rounds 3 --timeline --arms 0,0x800000 --names w6,w8
done
Q="--steps 40 --warmup 10 --no-cpu-baseline --no-torch-gpu-baseline --no-pmc --no-secondary --no-train"
for cfg in cfg2_dense_ri cfg2_ri; do
    timeout 200 python bench.py --config $cfg $Q > $OUT/n_bench_${cfg}.json 2> $OUT/n_bench_${cfg}.err; echo "[bench $cfg rc=$?]"
    python -c "
import json; d=json.loads(open('$OUT/n_bench_${cfg}.json').read().strip().splitlines()[-1])
print(round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,1), 'Mrays/s', d['config'].get('mlp_precision'), d['config'].get('table_precision'), d['config'].get('encoder_precision'), {k: (round(v,4) if isinstance(v,float) else v) for k,v in d['roofline'].items() if k in ('kernel_ms','frac','frac_executed')}, d['frame_timeline_ms'], d.get('parity_ok'))"
done
} > $OUT/r4_n.log 2>&1
cat $OUT/r4_n.log
