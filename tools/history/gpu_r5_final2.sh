#!/bin/bash
# round 5, the bench line again at HEAD (bench_generator.py now measures the product default -- producers replayed as hipGraphs -- with the eager number beside it)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > $OUT/r5_bench_final2.json 2> $OUT/r5_bench_final2.err; echo "[bench rc=$?]"; python -c "
import json; d=json.loads(open('$OUT/r5_bench_final2.json').read().strip().splitlines()[-1])
print({k: d[k] for k in list(d)[:12]})
print('roofline', {k: d['roofline'].get(k) for k in ('kernel','frac','kernel_ms','frac_executed','traffic','achieved')}); print('parity_ok', d.get('parity_ok'))
print('timeline', d.get('frame_timeline_ms')); sec = d.get('secondary') or {}
g = sec.get('generator_forward') or {}
print('generator', {k: g.get(k) for k in ('value','ms_per_step','rays_per_s_vs_renderer_alone','graphed','error')})
for k in ('recomputed_every_frame','recomputed_every_frame_eager_producers','use_cached_backbone'):
    print('  ', k, g.get(k))
print('train', (d.get('train') or {}).get('ms_per_step'))"
tail -3 $OUT/r5_bench_final2.err | cut -c1-300
timeout 300 python -m pytest tests/test_gpu_producers.py -q -x -m gpu -k "whole_generator" 2>&1 | tail -2
