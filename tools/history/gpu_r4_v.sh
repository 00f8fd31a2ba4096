#!/bin/bash
# round 4, call V: the default bench line at HEAD (four caller streams; roofline / timeline / secondary from the one-frame child)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python bench.py > $OUT/r4_bench_final.json 2> $OUT/r4_bench_final.err; echo "[bench rc=$?]"; python -c "
import json; d=json.loads(open('$OUT/r4_bench_final.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','ms_per_step','parity_ok','value_one_frame_in_flight','ms_per_step_one_frame_in_flight','latency_ms_per_frame_with_frames_overlapping','value_cfg2_wide_framing','ms_per_step_cfg2_wide_framing','value_cfg2_wide_framing_one_frame_in_flight','one_frame_in_flight_child') if k in d})
print('roofline', {k: v for k, v in d['roofline'].items() if k in ('frac','kernel_ms','frac_executed','traffic','kernel_ms_with_frames_overlapping')})
print('config', {k: d['config'].get(k) for k in ('valid_samples','mlp_precision','caller_streams','workspace_bytes','token_capacity')})
print('timeline', d.get('frame_timeline_ms')); print('train', (d.get('train') or {}).get('ms_per_step')); print('cpu', d.get('cpu_baseline', {}).get('value'))
print('secondary', json.dumps({k: ({kk: vv for kk, vv in v.items() if kk != 'frame_timeline_ms'} if isinstance(v, dict) else v) for k, v in d.get('secondary', {}).items() if k != 'mlp_kernel_alone'})[:1500])"
tail -3 $OUT/r4_bench_final.err | cut -c1-300
