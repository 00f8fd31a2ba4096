#!/bin/bash
# round 5, call K: the 70 us hole in front of the warp (rocprofv3 timeline of round 4): with / without the aux stream, with the joins in front of the warp removed (timing only)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python tools/frame_ab.py --config cfg2_dense_ri --arms 0,0,0x10000000 --names default,noaux,nojoin --opts ";aux_stream=False;" --timeline --rounds 3 > $OUT/r5k_frame_ab.txt 2>&1; echo "[frame_ab rc=$?]"
grep "^\[arm\|^\[bits\|^\[timeline\|Error" $OUT/r5k_frame_ab.txt | cut -c1-260
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/r5k_trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc --no-train > $OUT/r5k_trace.log 2>&1; echo "[trace rc=$?]"
DB=$(find $OUT/r5k_trace -name "*.db" | head -1); [ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB 2>/dev/null | head -90 | cut -c1-150 > $OUT/r5k_timeline.txt; head -80 $OUT/r5k_timeline.txt
rm -rf $OUT/r5k_trace
