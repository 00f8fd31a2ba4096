#!/bin/bash
# round 6, call R: two ray-side experiments of the first phase against the product, interleaved in one process (frames compared bit for bit by frame_ab):
# SHERF_EXPERIMENT bit 13 = the compaction with sixteen lanes per ray (four rays per wave), bit 14 = the list search one pipeline stage deeper (87 registers: five
# workgroups per CU), bits 14 + 15 = the same held to 80 registers (six per CU), debug 0x400000 = four search workgroups per CU
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python tools/frame_ab.py --config cfg2_dense_ri --arms 0,0,0,0,0x400000,0 --exps 0,8192,16384,49152,16384,24576 --names base,quad,deep5,deep6t,deep4,deep5quad --timeline --rounds 5 > $OUT/r6r_frame_ab.log 2>&1
echo "[ab rc=$?]"; grep "^\[timeline\|^\[arm\|^\[bits" $OUT/r6r_frame_ab.log | cut -c1-330
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc --no-train"
SHERF_EXPERIMENT=24576 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r -o trace -- $B > $OUT/prof_r.log 2>&1; echo "[rocprof rc=$?]"
DB=$(find $OUT/prof_r -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 40 > $OUT/r6r_prof_stats.txt; grep -i "cand_\|compact\|warp_geom" $OUT/r6r_prof_stats.txt | cut -c1-170
rm -rf $OUT/prof_r
