#!/bin/bash
# round 4, call H: the single-product sparse convolutions after the fix (direct fold diagnostic, layer-by-layer encoder diagnostic), the
# frame with auto free to pick them (bench, both framings), ISA-level regression: GPU tests of the touched paths
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 120 python tools/sconv_fold_diag.py > $OUT/h_sconv_fold_diag.txt 2>&1; echo "[fold diag rc=$?]"; grep -v amdgpu $OUT/h_sconv_fold_diag.txt | cut -c1-200 | tail -6
timeout 120 python tools/enc_sp_diag.py tiny_ri > $OUT/h_enc_diag.log 2>&1; echo "[enc diag rc=$?]"; grep -v "^/opt\|Warning" $OUT/h_enc_diag.log | tail -7 | cut -c1-250
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-pmc --no-secondary"
for cfg in cfg2_dense_ri cfg2_ri; do
  for enc in f16x3 auto; do
    E=""; [ $enc = f16x3 ] && E="--precision f16 --encoder-precision f16x3"
    timeout 120 python bench.py --config $cfg $Q $E > $OUT/h_bench_${cfg}_$enc.json 2> $OUT/h_bench_${cfg}_$enc.err; echo "[bench $cfg enc=$enc rc=$?]"
    python -c "
import json; d=json.loads(open('$OUT/h_bench_${cfg}_$enc.json').read().strip().splitlines()[-1])
print(round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,1), 'Mrays/s', d['config'].get('mlp_precision'), d['config'].get('table_precision'), d['config'].get('encoder_precision'), {k: (round(v,4) if isinstance(v,float) else v) for k,v in d['roofline'].items() if k in ('kernel_ms','frac','frac_executed')}, d['frame_timeline_ms'], (d['config'].get('mlp_precision_auto') or {}).get('errors_vs_reference_config'))"
  done
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -x -k "auto_precision or encoder or full_size_frame_properties" > $OUT/h_pytest.log 2>&1; echo "[pytest rc=$?]"; tail -4 $OUT/h_pytest.log | cut -c1-300; grep "^FAILED\|^ERROR\|^E  " $OUT/h_pytest.log | cut -c1-300 | head
