#!/bin/bash
# round 5, call Y: the forward gather with unconditional row loads (SHERF_EXPERIMENT bit 9) against the shipped kernel, both framings, bits + timeline
# (as it was run; the variant lost and its code -- bit 9 -- was removed again: DESIGN 9.27, profiles/r05_call_y_*)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for C in cfg2_dense_ri cfg2_ri; do
timeout 400 python tools/frame_ab.py --config $C --arms 0,0 --names shipped,uncond_loads --exps 0,512 --timeline --rounds 3 > $OUT/r5y_ab_$C.txt 2>&1; echo "[ab $C rc=$?]"
grep "^\[bits\]\|^\[arm\]\|^\[timeline\]" $OUT/r5y_ab_$C.txt | cut -c1-330
done
