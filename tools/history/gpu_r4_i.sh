#!/bin/bash
# round 4, call I: ray-side changes (candidate records with x_s, record prefetch, vector loads in the warp, early exit in the compaction):
# bit-exact ids / order on the hardware, frame time in both framings, kernel trace of the dense frame
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -x -k "mask_and_nearest or warp_matches or ragged or deterministic or margin_protocol or no_valid" > $OUT/i_pytest.log 2>&1; echo "[pytest rc=$?]"; tail -3 $OUT/i_pytest.log | cut -c1-300; grep "^FAILED\|^ERROR\|^E  " $OUT/i_pytest.log | cut -c1-300 | head
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-pmc --no-secondary"
for cfg in cfg2_dense_ri cfg2_ri; do
    timeout 120 python bench.py --config $cfg $Q > $OUT/i_bench_${cfg}.json 2> $OUT/i_bench_${cfg}.err; echo "[bench $cfg rc=$?]"
    python -c "
import json; d=json.loads(open('$OUT/i_bench_${cfg}.json').read().strip().splitlines()[-1])
print(round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,1), 'Mrays/s', d['config'].get('mlp_precision'), d['config'].get('table_precision'), d['config'].get('encoder_precision'), {k: (round(v,4) if isinstance(v,float) else v) for k,v in d['roofline'].items() if k in ('kernel_ms','frac','frac_executed')}, d['frame_timeline_ms'])"
done
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/i_prof -o trace -- $B > $OUT/i_prof.log 2>&1; echo "[rocprof rc=$?]"
DB=$(find $OUT/i_prof -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 45 > $OUT/i_prof_stats.txt; python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB > $OUT/i_prof_timeline.txt 2>&1; head -24 $OUT/i_prof_stats.txt | cut -c1-150
find $OUT/i_prof -name "*.db" -size +20M -delete
