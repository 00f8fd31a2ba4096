#!/bin/bash
# round 3, call A: the new parity protocol on the hardware (whole-frame fp64 truth, reference-init fixture), the default bench
# (cfg2_ri, precision auto), the adversarial workload beside it, the in-kernel timeline of the single-product MLP kernel, kernel trace
# and a counter pass for the MLP kernel.
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -s > $OUT/a_pytest_parity.log 2>&1; echo "[pytest parity rc=$?]"
tail -4 $OUT/a_pytest_parity.log | cut -c1-300; grep "^FAILED\|^ERROR\|verdict\|auto ->" $OUT/a_pytest_parity.log | cut -c1-300 | head -20
timeout 900 python bench.py > $OUT/a_bench.json 2> $OUT/a_bench.err; echo "[bench rc=$?]"; tail -3 $OUT/a_bench.err | cut -c1-300
python - <<PY
import json
try:
    d=json.loads(open('$OUT/a_bench.json').read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ('value','ms_per_step','value_dense','ms_per_step_dense','parity_ok','dtype')})
    print('roofline', d.get('roofline')); print('timeline', d.get('frame_timeline_ms')); print('cpu', d.get('cpu_baseline')); print('torch', d.get('torch_gpu_baseline'))
    print('auto', d['config'].get('mlp_precision_auto')); print((d.get('parity') or {}).get('table')); print('secondary', json.dumps(d.get('secondary'))[:2500])
except Exception as e: print('bench parse failed', e)
PY
timeout 600 python bench.py --config cfg2 --precision f16x3 --no-secondary --no-cpu-baseline > $OUT/a_bench_adversarial.json 2> $OUT/a_bench_adversarial.err; echo "[bench adversarial rc=$?]"
python - <<PY
import json
try:
    d=json.loads(open('$OUT/a_bench_adversarial.json').read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ('value','ms_per_step','parity_ok')}); print('roofline', d.get('roofline')); print((d.get('parity') or {}).get('table'))
except Exception as e: print('parse failed', e)
PY
timeout 300 python tools/mlp_trace.py --precision f16 --out $OUT/a_mlp_trace_f16.json > $OUT/a_mlp_trace_f16.log 2>&1; echo "[mlp trace f16 rc=$?]"; cat $OUT/a_mlp_trace_f16.log | cut -c1-260 | tail -25
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/a_prof -o trace -- $B > $OUT/a_prof.log 2>&1; echo "[rocprof rc=$?]"
DB=$(find $OUT/a_prof -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 45 > $OUT/a_prof_stats.txt; python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB > $OUT/a_prof_timeline.txt 2>&1; head -16 $OUT/a_prof_stats.txt | cut -c1-150
find $OUT/a_prof -name "*.db" -size +20M -delete
rocprofv3 -L 2>/dev/null | grep -i "mfma\|SQ_BUSY_CY\|GRBM_GUI\|SQ_WAVE_CYCLES\|SQ_INSTS_VALU \|SQ_ACTIVE_INST_VALU\|SQ_WAIT_INST_ANY\|SQ_WAIT_ANY" | cut -c1-160 | sort -u | head -40 > $OUT/a_counters_available.txt; wc -l $OUT/a_counters_available.txt
for P in f16 f16x3; do
  C="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --precision $P --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc"
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 -d $OUT/a_pmc_$P -o pmc -- $C > $OUT/a_pmc_$P.log 2>&1; echo "[pmc $P rc=$?]"
  DB=$(find $OUT/a_pmc_$P -name "*.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_query.py $DB nerf_mlp gather_tokens sample_nn > $OUT/a_pmc_$P.txt 2>&1; cat $OUT/a_pmc_$P.txt | cut -c1-120 | head -40
  find $OUT/a_pmc_$P -name "*.db" -size +20M -delete
done
