#!/bin/bash
# uncontended per-kernel durations: rocprofv3 kernel trace with the ray side serialised behind the encoder
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf $OUT/tr; mkdir -p $OUT/tr
SHERF_MAIN_AFTER_LAYER=13 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/tr -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/tr.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $OUT/tr/t_results.db > $OUT/timeline_serial.txt
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $OUT/tr/t_results.db 13 45 > $OUT/kernel_stats_serial.txt
rm -rf $OUT/tr
