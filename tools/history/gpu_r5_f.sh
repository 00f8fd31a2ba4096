#!/bin/bash
# round 5, call F: new GPU tests (launch forms, full-size backbone, full-size backward vs the float64 truth), TriPlaneGenerator.forward at full size, a first look at the new bench line
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_glue.py tests/test_gpu_producers.py -q -x -m gpu -k "launch_forms or full_size or whole_generator" > $OUT/r5f_pytest_forms_backbone.txt 2>&1; echo "[pytest forms/backbone rc=$?]"; tail -4 $OUT/r5f_pytest_forms_backbone.txt
timeout 300 python bench_generator.py --steps 20 --warmup 5 > $OUT/r5f_bench_generator.json 2> $OUT/r5f_bench_generator.err; echo "[bench_generator rc=$?]"; tail -c 2500 $OUT/r5f_bench_generator.json; tail -3 $OUT/r5f_bench_generator.err
timeout 900 python -m pytest tests/test_gpu_backward.py -q -x -m gpu -s -k "full_size_backward" > $OUT/r5f_pytest_backward_truth64.txt 2>&1; echo "[pytest backward rc=$?]"; grep -v "^$" $OUT/r5f_pytest_backward_truth64.txt | tail -40
timeout 600 python bench.py --steps 20 --warmup 5 --no-train --no-pmc > $OUT/r5f_bench.json 2> $OUT/r5f_bench.err; echo "[bench rc=$?]"; python - <<'PY'
import json
try:
    d = json.loads([l for l in open('/root/repo/gpurun_out/r5f_bench.json') if l.startswith('{')][-1])
    keep = {k: d.get(k) for k in list(d)[:12]}
    keep['roofline'] = {k: d['roofline'].get(k) for k in ('kernel', 'frac', 'kernel_ms', 'frac_executed')} if isinstance(d.get('roofline'), dict) else d.get('roofline')
    keep['timeline'] = d.get('frame_timeline_ms'); keep['parity_ok'] = d.get('parity_ok')
    sec = d.get('secondary', {})
    keep['fresh'] = sec.get('fresh_inputs'); keep['mlp'] = {k: (v.get('kernel_ms'), v.get('frac')) for k, v in (sec.get('mlp_kernel_alone') or {}).items() if isinstance(v, dict)}
    keep['generator_forward'] = {k: sec.get('generator_forward', {}).get(k) for k in ('value', 'ms_per_step', 'recomputed_every_frame', 'use_cached_backbone', 'error')}
    print(json.dumps(keep, indent=1)[:6000])
except Exception as ex:
    print('no line:', ex)
PY
tail -5 $OUT/r5f_bench.err
