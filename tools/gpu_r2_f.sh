#!/bin/bash
# round 2, call F: trimmed sampler, MFMA backward GEMMs, fma_mix variant, CU-masked encoder streams
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python tools/mlp_trace.py > $OUT/mlp_trace.log 2>&1; echo "[trace rc=$?]"; grep "^\[lib\|^\[bf16\|^\[roof\|^\[trace" $OUT/mlp_trace.log | cut -c1-200
SHERF_HIP_LIB=$GRAFT_REPO_ROOT/sherf_amd/libsherf_hip_mix.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -s -k "per_sample_sigma or end_to_end or mask_and_nearest" > $OUT/mix.log 2>&1; echo "[mix parity rc=$?]"; grep "rel-to-max\|PSNR\|passed\|failed" $OUT/mix.log | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_producers.py tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -s -k "backward or gemm or producers or generator or loss or resnet or stylegan or mask_and_nearest or stage or margin_protocol or dataset or eval_mode" > $OUT/pytest_sel.log 2>&1; echo "[pytest selected rc=$?]"; tail -4 $OUT/pytest_sel.log; grep "FAILED\|conditioning\|Error" $OUT/pytest_sel.log | cut -c1-300 | head -20
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc"
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), d['frame_timeline_ms'])"; }
$B 2>/dev/null | grep '"metric"' | pr default
SHERF_ENCODER_CU_MASK=ffffffff $B 2>/dev/null | grep '"metric"' | pr mask_lo32
SHERF_ENCODER_CU_MASK=ffffffff,ffffffff $B 2>/dev/null | grep '"metric"' | pr mask_lo64
SHERF_ENCODER_CU_MASK=11111111,11111111,11111111,11111111,11111111,11111111,11111111,11111111 $B 2>/dev/null | grep '"metric"' | pr mask_every4th
SHERF_ENCODER_CU_MASK=01010101,01010101,01010101,01010101,01010101,01010101,01010101,01010101 $B 2>/dev/null | grep '"metric"' | pr mask_every8th
$B --exact-grids 2>/dev/null | grep '"metric"' | pr exact_grids
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_f -o trace -- $B > $OUT/prof_f.log 2>&1; echo "[rocprof rc=$?]"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_f_train -o trace -- python $GRAFT_REPO_ROOT/bench_train.py --steps 3 --warmup 1 > $OUT/prof_f_train.log 2>&1; echo "[rocprof train rc=$?]"; grep '"metric"' $OUT/prof_f_train.log | cut -c1-500
cd $GRAFT_REPO_ROOT
for d in prof_f prof_f_train; do DB=$(find $OUT/$d -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB 0 40 > $OUT/${d}_stats.txt; echo "== $d"; head -16 $OUT/${d}_stats.txt | cut -c1-150; find $OUT/$d -name "*.db" -size +20M -delete; done
