#!/bin/bash
# round 4, final evidence run: the default bench line (parity, PMC traffic, cpu_baseline, train, secondary), then the rocprofv3 kernel trace +
# per-stream timeline of the same command (the whole GPU suite at HEAD: call O, profiles/r04_pytest_gpu_head.txt)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > $OUT/r4_bench_final.json 2> $OUT/r4_bench_final.err; echo "[bench rc=$?]"; python -c "
import json; d=json.loads(open('$OUT/r4_bench_final.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','ms_per_step','roofline','cpu_baseline','parity_ok','value_cfg2_wide_framing','ms_per_step_cfg2_wide_framing') if k in d})
print('config', {k: d['config'].get(k) for k in ('workload','valid_samples','mlp_precision','table_precision','encoder_precision','workspace_bytes','token_capacity','sampler_capacity')})
print('timeline', d.get('frame_timeline_ms')); print('parity', json.dumps(d.get('parity'))[:1200]); print('train', json.dumps(d.get('train'))[:900]); print('secondary', json.dumps(d.get('secondary'))[:2500]); print('torch', d.get('torch_gpu_baseline'))"
tail -5 $OUT/r4_bench_final.err | cut -c1-300
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc --no-train"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/r4_prof_final -o trace -- $B > $OUT/r4_prof_final.log 2>&1; echo "[rocprof rc=$?]"
DB=$(find $OUT/r4_prof_final -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 50 > $OUT/r4_prof_final_stats.txt; python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB > $OUT/r4_prof_final_timeline.txt 2>&1; head -16 $OUT/r4_prof_final_stats.txt | cut -c1-140
find $OUT/r4_prof_final -name "*.db" -size +20M -delete
