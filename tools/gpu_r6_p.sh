#!/bin/bash
# round 6, call P: the warp's nearest-T-vertex search walking its (up to nine) row segments one after the other (product) against the walk over the concatenated
# segments with an eight-way select per point (libsherf_hip_nnold.so): one frame_ab process per library on one box, twice; frames compared bit for bit
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for tag in product nnold product nnold; do
  lib=sherf_amd/libsherf_hip_$tag.so; [ $tag = product ] && lib=sherf_amd/libsherf_hip.so
  SHERF_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout 600 python tools/frame_ab.py --config cfg2_dense_ri --arms 0 --names $tag --timeline --rounds 4 --dump $OUT/r6p_frame_$tag.pt > $OUT/r6p_frame_ab_$tag.log 2>&1
  echo "[$tag rc=$?]"; grep "^\[timeline\|^\[arm" $OUT/r6p_frame_ab_$tag.log | cut -c1-330
done
python - <<'PY'
import torch, os
out = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out'
a, b = torch.load(out + '/r6p_frame_product.pt'), torch.load(out + '/r6p_frame_nnold.pt')
print('frames identical:', all(torch.equal(x, y) for x, y in zip(a, b)))
PY
rm -f $OUT/r6p_frame_*.pt
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc --no-train"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_p -o trace -- $B > $OUT/prof_p.log 2>&1; echo "[rocprof rc=$?]"
DB=$(find $OUT/prof_p -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 12 > $OUT/r6p_prof_stats.txt; head -14 $OUT/r6p_prof_stats.txt | cut -c1-150
rm -rf $OUT/prof_p
