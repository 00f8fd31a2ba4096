#!/bin/bash
# round 4, call A: the two-launch MLP on hardware (bit-identity tests, A/B timings against the one-launch kernel incl. build variants, a
# bitwise stress test under side-stream load), the single-product encoder diagnostic (folds sliced to their valid rows), bench lines in both
# framings and both MLP forms, kernel trace of the default bench.
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_glue.py -m gpu -q --no-header -p no:cacheprovider -x > $OUT/a_pytest_glue.log 2>&1; echo "[pytest glue rc=$?]"; tail -3 $OUT/a_pytest_glue.log | cut -c1-300; grep "^FAILED\|^ERROR\|^E  " $OUT/a_pytest_glue.log | cut -c1-300 | head
timeout 200 python tools/mlp_ab.py --stress 200 --rounds 3 --out $OUT/a_mlp_ab_f16.json > $OUT/a_mlp_ab_f16.log 2>&1; echo "[ab f16 rc=$?]"; grep "^\[" $OUT/a_mlp_ab_f16.log | cut -c1-220
timeout 200 python tools/mlp_ab.py --config cfg2_dense_ri --rounds 2 --out $OUT/a_mlp_ab_dense.json > $OUT/a_mlp_ab_dense.log 2>&1; echo "[ab dense rc=$?]"; grep "^\[" $OUT/a_mlp_ab_dense.log | cut -c1-220
timeout 200 python tools/mlp_ab.py --precision f16x3 --config cfg2 --rounds 2 --stress 100 --out $OUT/a_mlp_ab_x3.json > $OUT/a_mlp_ab_x3.log 2>&1; echo "[ab x3 rc=$?]"; grep "^\[" $OUT/a_mlp_ab_x3.log | cut -c1-220
timeout 120 python tools/enc_sp_diag.py tiny_ri > $OUT/a_enc_diag.log 2>&1; echo "[enc diag rc=$?]"; grep -v "^/opt\|Warning" $OUT/a_enc_diag.log | tail -22 | cut -c1-250
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-pmc --no-secondary"
for form in 0 1; do
  SHERF_MLP_SPLIT=$form timeout 120 python bench.py --config cfg2_ri $Q > $OUT/a_bench_cfg2ri_split$form.json 2> $OUT/a_bench_cfg2ri_split$form.err; echo "[bench cfg2_ri split=$form rc=$?]"
  python -c "
import json; d=json.loads(open('$OUT/a_bench_cfg2ri_split$form.json').read().strip().splitlines()[-1])
print(round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,1), 'Mrays/s', {k: (round(v,4) if isinstance(v,float) else v) for k,v in d['roofline'].items() if k in ('kernel_ms','frac','frac_executed')}, d['frame_timeline_ms'])"
done
timeout 600 python bench.py > $OUT/a_bench.json 2> $OUT/a_bench.err; echo "[bench default rc=$?]"; python -c "
import json; d=json.loads(open('$OUT/a_bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','ms_per_step','roofline','cpu_baseline','value_cfg2_wide_framing') if k in d})
print('timeline', d.get('frame_timeline_ms')); print('parity_ok', d.get('parity_ok'), json.dumps(d.get('parity'))[:900]); print('secondary', json.dumps(d.get('secondary'))[:2500]); print('torch', d.get('torch_gpu_baseline'))"
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/a_prof -o trace -- $B > $OUT/a_prof.log 2>&1; echo "[rocprof rc=$?]"
DB=$(find $OUT/a_prof -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 45 > $OUT/a_prof_stats.txt; python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB > $OUT/a_prof_timeline.txt 2>&1; head -16 $OUT/a_prof_stats.txt | cut -c1-150
find $OUT/a_prof -name "*.db" -size +20M -delete
