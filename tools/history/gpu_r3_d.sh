#!/bin/bash
# round 3, call D: single-product sparse convolutions under the calibrated configuration, capped candidate search; gather counters
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -s -k "auto or sparse_voxel or mask_and or (full_size and ri)" > $OUT/d_pytest.log 2>&1; echo "[pytest rc=$?]"
tail -3 $OUT/d_pytest.log | cut -c1-300; grep "^FAILED\|^ERROR\|verdict\|auto ->\|clean " $OUT/d_pytest.log | cut -c1-300 | head -20
Q="--steps 40 --warmup 10 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc"
run() { env $1 timeout 300 python bench.py $Q $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,1), 'Mrays/s mlp', round(d['roofline']['kernel_ms'],4), d['config'].get('mlp_precision'), d['config'].get('table_precision'), d['config'].get('encoder_precision'), d.get('frame_timeline_ms'))"; }
run X=0 ""
run X=0 "--precision f16 --encoder-precision f16x3"
run X=0 "--precision f16 --encoder-precision f16x3 --table-precision f32"
run SHERF_DEBUG=512 ""
run X=0 "--config cfg2_dense_ri"
run X=0 "--config cfg3_ri"
timeout 600 python bench.py --no-cpu-baseline > $OUT/d_bench.json 2> $OUT/d_bench.err; echo "[bench rc=$?]"; tail -2 $OUT/d_bench.err | cut -c1-300
python - <<PY
import json
try:
    d=json.loads(open('$OUT/d_bench.json').read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ('value','ms_per_step','value_dense','ms_per_step_dense','parity_ok','dtype')})
    print('roofline', {k: d['roofline'][k] for k in ('frac','kernel_ms','traffic')}); print('timeline', d.get('frame_timeline_ms'))
    print('auto', d['config'].get('mlp_precision_auto')); print((d.get('parity') or {}).get('table')); print('plain', {k: (d.get('parity') or {}).get('samples', {}).get(k) for k in ('sigma_rel_max','rgb_rel_max')})
except Exception as e: print('bench parse failed', e)
PY
rocprofv3 -L 2>/dev/null | grep -o "\b\(TA\|TCP\|TCC\|TD\)_[A-Z0-9_a-z\[\]]*" | sort -u > $OUT/d_counters_mem.txt; wc -l $OUT/d_counters_mem.txt; grep -c . $OUT/d_counters_mem.txt
cd /tmp
C="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc"
for SET in "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  TAG=$(echo $SET | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $SET -d $OUT/d_pmc_$TAG -o pmc -- $C > $OUT/d_pmc_$TAG.log 2>&1; echo "[pmc $TAG rc=$?]"
  DB=$(find $OUT/d_pmc_$TAG -name "*.db" | head -1)
  [ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/pmc_query.py $DB gather_tokens nerf_mlp cand_search 2>&1 | cut -c1-120 | grep -v "^# pmc" | head -30
  find $OUT/d_pmc_$TAG -name "*.db" -size +20M -delete
done
