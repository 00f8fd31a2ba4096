#!/bin/bash
# round 6, call J: -fno-slp-vectorize for gather.hip / sample.hip / svox.hip (hipcc's SLP pass turns the fp16 taps' `acc += w * (float)v` into 8 v_cvt_f32_f16 + 4
# v_pk_fma_f32 where 8 v_fma_mix_f32 do; 110 -> 96 VGPRs in the gather): one frame_ab process per library (same box, back to back), timeline + bits of the rendered frame
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for tag in product gnoslp snoslp vnoslp allnoslp product; do
  lib=sherf_amd/libsherf_hip_$tag.so; [ $tag = product ] && lib=sherf_amd/libsherf_hip.so
  SHERF_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout 600 python tools/frame_ab.py --config cfg2_dense_ri --arms 0 --names $tag --timeline --rounds 4 --dump $OUT/r6j_frame_$tag.pt > $OUT/r6j_frame_ab_$tag.log 2>&1
  echo "[$tag rc=$?]"; grep "^\[timeline\|^\[arm" $OUT/r6j_frame_ab_$tag.log | cut -c1-330
done
python - <<'PY'
import torch, glob, os
out = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out'
ref = torch.load(out + '/r6j_frame_product.pt')
for f in sorted(glob.glob(out + '/r6j_frame_*.pt')):
    t = torch.load(f)
    print(os.path.basename(f), 'identical to product:', all(torch.equal(a, b) for a, b in zip(t, ref)), 'max |d rgb|', float((t[0] - ref[0]).abs().max()))
PY
rm -f $OUT/r6j_frame_*.pt
