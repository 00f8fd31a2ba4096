#!/bin/bash
# round 4, call U: the default bench with four caller streams (value) + one frame in flight (roofline, timeline); stream-count sweep;
# rocprofv3 kernel traces of the default command and of --streams 1
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
Q="--steps 48 --warmup 12 --no-cpu-baseline --no-torch-gpu-baseline --no-pmc --no-secondary --no-train"
{
for cfg in cfg2_dense_ri cfg2_ri; do
  for n in 1 4 5 8 4 1; do
    timeout 200 python bench.py --config $cfg --streams $n $Q > $OUT/u_bench.json 2> $OUT/u_bench.err; rc=$?
    python -c "
import json; d=json.loads(open('$OUT/u_bench.json').read().strip().splitlines()[-1])
print('$cfg streams $n rc=$rc:', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,1), 'Mrays/s', 'one-in-flight', d.get('ms_per_step_one_frame_in_flight'), 'latency', d.get('latency_ms_per_frame_with_frames_overlapping'), 'mlp', round(d['roofline']['kernel_ms'],4), d['roofline'].get('kernel_ms_with_frames_overlapping'))"
  done
done
} > $OUT/r4_u.log 2>&1
cat $OUT/r4_u.log
timeout 900 python bench.py > $OUT/r4_bench_final.json 2> $OUT/r4_bench_final.err; echo "[bench rc=$?]"; python -c "
import json; d=json.loads(open('$OUT/r4_bench_final.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','ms_per_step','parity_ok','value_one_frame_in_flight','ms_per_step_one_frame_in_flight','latency_ms_per_frame_with_frames_overlapping','value_cfg2_wide_framing','ms_per_step_cfg2_wide_framing','value_cfg2_wide_framing_one_frame_in_flight') if k in d})
print('roofline', {k: v for k, v in d['roofline'].items() if k in ('frac','kernel_ms','frac_executed','traffic','kernel_ms_with_frames_overlapping')})
print('config', {k: d['config'].get(k) for k in ('valid_samples','mlp_precision','caller_streams','workspace_bytes','token_capacity')})
print('timeline', d.get('frame_timeline_ms')); print('train', (d.get('train') or {}).get('ms_per_step')); print('cpu', d.get('cpu_baseline', {}).get('value'))
print('secondary', json.dumps({k: {kk: vv for kk, vv in v.items() if kk != 'frame_timeline_ms'} for k, v in d.get('secondary', {}).items() if k != 'mlp_kernel_alone'})[:1500])"
tail -3 $OUT/r4_bench_final.err | cut -c1-300
cd /tmp
for n in 4 1; do
B="python $GRAFT_REPO_ROOT/bench.py --streams $n --steps 20 --warmup 8 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc --no-train"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/r4_prof_s$n -o trace -- $B > $OUT/r4_prof_s$n.log 2>&1; echo "[rocprof streams=$n rc=$?]"
DB=$(find $OUT/r4_prof_s$n -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 50 > $OUT/r4_prof_s${n}_stats.txt; python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB > $OUT/r4_prof_s${n}_timeline.txt 2>&1; head -8 $OUT/r4_prof_s${n}_stats.txt | cut -c1-140
find $OUT/r4_prof_s$n -name "*.db" -size +20M -delete
done
