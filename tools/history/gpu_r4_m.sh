#!/bin/bash
# round 4, call M: with the shorter list search -- how much of the chip should it hold (persistent workgroups per CU), should the ray side
# start behind the encoder's first layers (main_after_layer), un-profiled HIP-event timelines of every arm
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
{
for cfg in cfg2_dense_ri cfg2_ri; do
  timeout 400 python tools/frame_ab.py --config $cfg --rounds 3 --timeline \
     --arms 0,0x600000,0x400000,0x300000,0,0,0,0x4000 --names w8,w6,w4,w3,after0,after2,after4,cellwalk \
     --opts ";;;;main_after_layer=0;main_after_layer=2;main_after_layer=4;"
done
} > $OUT/r4_m.log 2>&1
cat $OUT/r4_m.log
