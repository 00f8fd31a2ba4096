#!/bin/bash
# round 3, call C: two-pass sampler + frames on two caller streams + adopted MLP variants
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -s -k "mask_and or warp or deterministic or ragged or rotation or auto or no_valid or full_size" > $OUT/c_pytest.log 2>&1; echo "[pytest rc=$?]"
tail -3 $OUT/c_pytest.log | cut -c1-300; grep "^FAILED\|^ERROR\|verdict\|auto ->" $OUT/c_pytest.log | cut -c1-260 | head -20
Q="--steps 40 --warmup 10 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc"
run() { env $1 timeout 300 python bench.py $Q $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,1), 'Mrays/s mlp', round(d['roofline']['kernel_ms'],4), d.get('frame_timeline_ms'))"; }
run X=0 "--streams 1"
run SHERF_DEBUG=512 "--streams 1"
run X=0 "--streams 2"
run SHERF_DEBUG=512 "--streams 2"
run X=0 "--streams 3"
run GPU_MAX_HW_QUEUES=4 "--streams 2"
run X=0 "--streams 2 --config cfg2_dense_ri"
run X=0 "--streams 1 --config cfg2_dense_ri"
run X=0 "--streams 2 --config cfg2 --precision f16x3"
timeout 600 python bench.py --no-cpu-baseline > $OUT/c_bench.json 2> $OUT/c_bench.err; echo "[bench rc=$?]"; tail -2 $OUT/c_bench.err | cut -c1-300
python - <<PY
import json
try:
    d=json.loads(open('$OUT/c_bench.json').read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ('value','ms_per_step','value_dense','ms_per_step_dense','parity_ok','dtype')})
    print('roofline', {k: d['roofline'][k] for k in ('frac','kernel_ms','traffic')}); print('timeline', d.get('frame_timeline_ms'))
    print((d.get('parity') or {}).get('table'))
    s=d.get('secondary', {}); print({k: (v.get('ms_per_frame'), v.get('mlp_precision')) for k, v in s.items() if isinstance(v, dict) and 'ms_per_frame' in v})
except Exception as e: print('bench parse failed', e)
PY
cd /tmp
for ST in 1 2; do
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 6 --streams $ST --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/c_prof$ST -o trace -- $B > $OUT/c_prof$ST.log 2>&1; echo "[rocprof streams=$ST rc=$?]"
DB=$(find $OUT/c_prof$ST -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 45 > $OUT/c_prof${ST}_stats.txt; python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB > $OUT/c_prof${ST}_timeline.txt 2>&1; head -14 $OUT/c_prof${ST}_stats.txt | cut -c1-150
find $OUT/c_prof$ST -name "*.db" -size +20M -delete
done
