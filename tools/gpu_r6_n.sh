#!/bin/bash
# round 6, call N: mlp_form='auto' (the fastest of the three bit-identical launch forms of the network, timed once per device and precision) -- the glue tests that
# compare the forms, then the default bench line
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_glue.py tests/test_gpu_parity.py -q -m gpu --no-header -p no:cacheprovider -x -k "launch_forms or schedule_switches or without_transformer or auto_precision or deterministic" > $OUT/r6n_pytest.log 2>&1
echo "[pytest rc=$?]"; tail -3 $OUT/r6n_pytest.log | cut -c1-300
timeout 1500 python bench.py --no-train > $OUT/r6n_bench.json 2> $OUT/r6n_bench.err; echo "[bench rc=$?]"; python -c "
import json; d=json.loads(open('$OUT/r6n_bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in list(d)[:8]})
print('roofline', {k: d['roofline'].get(k) for k in ('kernel','frac','kernel_ms','traffic')}); print('parity_ok', d.get('parity_ok'))
print('form', d['config'].get('mlp_form'), d['config'].get('mlp_form_auto')); print('timeline', d.get('frame_timeline_ms'))
m=d['secondary']['mlp_kernel_alone']; print({k:(round(v['kernel_ms'],4), round(v.get('frac',0),3)) for k,v in m.items() if isinstance(v,dict) and 'kernel_ms' in v})"
