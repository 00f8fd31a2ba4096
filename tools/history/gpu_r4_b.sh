#!/bin/bash
# round 4, call B: where does a decoder step go?  Ablation builds of the MLP kernels (no weight DMA / no barriers / no per-step A-fragment
# reads / none of the three), the A-fragment window against round 3's one-block prefetch, the static tokens kernel; both launch forms.
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python tools/mlp_ab.py --rounds 3 --out $OUT/b_mlp_ab_f16.json > $OUT/b_mlp_ab_f16.log 2>&1; echo "[ab f16 cfg2_ri rc=$?]"; grep "^\[" $OUT/b_mlp_ab_f16.log | cut -c1-220
timeout 300 python tools/mlp_ab.py --config cfg2_dense_ri --rounds 2 --stress 100 --out $OUT/b_mlp_ab_dense.json > $OUT/b_mlp_ab_dense.log 2>&1; echo "[ab dense rc=$?]"; grep "^\[" $OUT/b_mlp_ab_dense.log | cut -c1-220
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-pmc --no-secondary"
for form in 0 1; do
  SHERF_MLP_SPLIT=$form timeout 120 python bench.py $Q > $OUT/b_bench_dense_split$form.json 2> $OUT/b_bench_dense_split$form.err; echo "[bench dense split=$form rc=$?]"
  python -c "
import json; d=json.loads(open('$OUT/b_bench_dense_split$form.json').read().strip().splitlines()[-1])
print(round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,1), 'Mrays/s', {k: (round(v,4) if isinstance(v,float) else v) for k,v in d['roofline'].items() if k in ('kernel_ms','frac','frac_executed')}, d['frame_timeline_ms'])"
done
