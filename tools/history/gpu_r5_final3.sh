#!/bin/bash
# round 5, SECOND evidence run at HEAD (after the training-step work and the first-phase experiments): the whole GPU suite, the default bench line,
# secondary incl. generator_forward and fresh_inputs), the rocprofv3 kernel trace + per-stream timeline of the same command, the training step's kernel trace
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -s > $OUT/r5_pytest_final3.log 2>&1; echo "[pytest rc=$?]"; tail -4 $OUT/r5_pytest_final3.log | cut -c1-300; grep "^FAILED\|^ERROR" $OUT/r5_pytest_final3.log | cut -c1-300 | head
timeout 900 python bench.py > $OUT/r5_bench_final3.json 2> $OUT/r5_bench_final3.err; echo "[bench rc=$?]"; python -c "
import json; d=json.loads(open('$OUT/r5_bench_final3.json').read().strip().splitlines()[-1])
print({k: d[k] for k in list(d)[:12]})
print('roofline', {k: d['roofline'].get(k) for k in ('kernel','frac','kernel_ms','frac_executed','traffic','achieved')}); print('cpu', d.get('cpu_baseline')); print('parity_ok', d.get('parity_ok'))
print('config', {k: d['config'].get(k) for k in ('workload','valid_samples','mlp_precision','table_precision','encoder_precision','workspace_bytes','caller_streams','inputs','exchange')})
print('timeline', d.get('frame_timeline_ms')); print('train', json.dumps(d.get('train'))[:700]); sec = d.get('secondary') or {}
print('mlp alone', {k: (round(v.get('kernel_ms', 0), 4), round(v.get('frac') or 0, 3)) for k, v in (sec.get('mlp_kernel_alone') or {}).items() if isinstance(v, dict)})
print('fresh', sec.get('fresh_inputs')); print('generator', json.dumps(sec.get('generator_forward'))[:1500])
print('others', {k: (v.get('ms_per_frame'), v.get('rays_per_s')) for k, v in sec.items() if isinstance(v, dict) and 'ms_per_frame' in v}); print('torch', d.get('torch_gpu_baseline'))"
tail -5 $OUT/r5_bench_final3.err | cut -c1-300
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc --no-train"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/r5_prof_final3 -o trace -- $B > $OUT/r5_prof_final3.log 2>&1; echo "[rocprof rc=$?]"
DB=$(find $OUT/r5_prof_final3 -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 50 > $OUT/r5_prof_final3_stats.txt; python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB > $OUT/r5_prof_final3_timeline.txt 2>&1; head -16 $OUT/r5_prof_final3_stats.txt | cut -c1-140
rm -rf $OUT/r5_prof_final3
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/r5_prof_train3 -o trace -- python $GRAFT_REPO_ROOT/bench_train.py --steps 3 --warmup 1 > $OUT/r5_prof_train3.log 2>&1; echo "[train prof rc=$?]"
DB=$(find $OUT/r5_prof_train3 -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 70 > $OUT/r5_prof_train3_stats.txt; head -12 $OUT/r5_prof_train3_stats.txt | cut -c1-140; rm -rf $OUT/r5_prof_train3
