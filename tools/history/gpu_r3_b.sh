#!/bin/bash
# round 3, call B: fp16 tables + end-to-end auto calibration on the hardware, MLP micro-variants (packed-f16 ReLU, no SLP, launch bounds),
# the gather split, default bench.
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -s -k "auto or eval_mode or full_size or margin or native" > $OUT/b_pytest.log 2>&1; echo "[pytest rc=$?]"
tail -3 $OUT/b_pytest.log | cut -c1-300; grep "^FAILED\|^ERROR\|verdict\|auto ->\|clean " $OUT/b_pytest.log | cut -c1-260 | head -30
timeout 300 python tools/mlp_trace.py --precision f16 --out $OUT/b_mlp_trace_f16.json > $OUT/b_mlp_trace_f16.log 2>&1; echo "[mlp trace rc=$?]"; grep "^\[lib\]\|^\[roofline\]\|^\[trace\]" $OUT/b_mlp_trace_f16.log | cut -c1-200
timeout 600 python bench.py --no-cpu-baseline > $OUT/b_bench.json 2> $OUT/b_bench.err; echo "[bench rc=$?]"; tail -2 $OUT/b_bench.err | cut -c1-300
python - <<PY
import json
try:
    d=json.loads(open('$OUT/b_bench.json').read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ('value','ms_per_step','value_dense','ms_per_step_dense','parity_ok','dtype')})
    print('roofline', {k: d['roofline'][k] for k in ('frac','kernel_ms','traffic')}); print('timeline', d.get('frame_timeline_ms'))
    print('auto', d['config'].get('mlp_precision_auto')); print((d.get('parity') or {}).get('table')); print('plain', {k: (d.get('parity') or {}).get('samples', {}).get(k) for k in ('sigma_rel_max','rgb_rel_max')})
    s=d.get('secondary', {}); print({k: (v.get('ms_per_frame'), v.get('mlp_precision')) for k, v in s.items() if isinstance(v, dict) and 'ms_per_frame' in v})
except Exception as e: print('bench parse failed', e)
PY
for V in "SHERF_GATHER_SPLIT=1" "SHERF_GATHER_BRANCHLESS=1" "SHERF_GATHER_BRANCHLESS=128"; do
  env $V timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', d['ms_per_step'], d.get('frame_timeline_ms'))"
done
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/b_prof -o trace -- $B > $OUT/b_prof.log 2>&1; echo "[rocprof rc=$?]"
DB=$(find $OUT/b_prof -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 45 > $OUT/b_prof_stats.txt; python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB > $OUT/b_prof_timeline.txt 2>&1; head -12 $OUT/b_prof_stats.txt | cut -c1-150
find $OUT/b_prof -name "*.db" -size +20M -delete
