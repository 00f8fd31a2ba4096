#!/bin/bash
# frame-driver iteration: GPU tests, timeline of one step, bench with and without the gather split
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -x --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 $OUT/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep '"metric"\|Error\|error' | tee $OUT/bench.log | cut -c1-260
SHERF_GATHER_SPLIT=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep '"metric"' | tee $OUT/bench_split.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('SPLIT: ms/step', d['ms_per_step'], 'mlp ms', d['roofline']['kernel_ms'])"
cd /tmp
rm -rf $OUT/tr; mkdir -p $OUT/tr
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/tr.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $OUT/tr/t_results.db > $OUT/timeline.txt
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $OUT/tr/t_results.db 13 45 > $OUT/kernel_stats.txt; cut -c1-150 $OUT/kernel_stats.txt | head -24
grep '"metric"' $OUT/tr.log | cut -c1-200
rm -rf $OUT/tr
