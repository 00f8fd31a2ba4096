#!/bin/bash
# final evidence run of a round (rounds 3-6): default bench (parity, PMC traffic, secondary lines), rocprofv3 kernel trace + timeline of the same command,
# training-step bench + its kernel trace, then the whole GPU suite.
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1800 python bench.py > $OUT/bench_final.json 2> $OUT/bench_final.err; echo "[bench rc=$?]"; python -c "
import json; d=json.loads(open('$OUT/bench_final.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','ms_per_step','roofline','cpu_baseline') if k in d})
print('timeline', d.get('frame_timeline_ms')); print('parity', json.dumps(d.get('parity'))[:1500]); print('secondary', json.dumps(d.get('secondary'))[:1800]); print('torch', d.get('torch_gpu_baseline'))"
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_final -o trace -- $B > $OUT/prof_final.log 2>&1; echo "[rocprof rc=$?]"
DB=$(find $OUT/prof_final -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 45 20 > $OUT/prof_final_stats.txt; python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB > $OUT/prof_final_timeline.txt 2>&1; head -14 $OUT/prof_final_stats.txt | cut -c1-140
find $OUT/prof_final -name "*.db" -size +20M -delete
cd $GRAFT_REPO_ROOT; timeout 300 python bench_train.py --steps 4 --warmup 2 --no-pmc > $OUT/train_final.json 2> $OUT/train_final.err; echo "[train rc=$?]"; cut -c1-1500 $OUT/train_final.json; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_final_train -o trace -- python $GRAFT_REPO_ROOT/bench_train.py --steps 3 --warmup 1 --no-pmc > $OUT/train_final.log 2>&1; echo "[train prof rc=$?]"
DB=$(find $OUT/prof_final_train -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 30 > $OUT/prof_final_train_stats.txt; find $OUT/prof_final_train -name "*.db" -size +20M -delete
cd $GRAFT_REPO_ROOT
timeout 2700 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -s > $OUT/pytest_final.log 2>&1; echo "[pytest rc=$?]"; tail -6 $OUT/pytest_final.log | cut -c1-300; grep "^FAILED\|^ERROR" $OUT/pytest_final.log | cut -c1-300 | head
