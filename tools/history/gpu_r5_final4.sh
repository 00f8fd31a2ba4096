#!/bin/bash
# round 5, last GPU call: the backward's GPU tests + the frame / parity modules at HEAD, then the default bench line again (its `train` entry is HEAD's step)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_glue.py -q -m gpu --no-header -p no:cacheprovider > $OUT/r5_pytest_final4.log 2>&1; echo "[pytest rc=$?]"; tail -3 $OUT/r5_pytest_final4.log | cut -c1-200
timeout 900 python bench.py > $OUT/r5_bench_final4.json 2> $OUT/r5_bench_final4.err; echo "[bench rc=$?]"; python -c "
import json; d=json.loads(open('$OUT/r5_bench_final4.json').read().strip().splitlines()[-1])
print({k: d[k] for k in list(d)[:12]})
print('roofline', {k: d['roofline'].get(k) for k in ('kernel','frac','kernel_ms','traffic')}); print('parity_ok', d.get('parity_ok'))
print('train', json.dumps(d.get('train'))[:400]); print('timeline', d.get('frame_timeline_ms'))"
