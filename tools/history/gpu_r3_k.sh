#!/bin/bash
# round 3, call K: after the final run -- the config-1 truth protocol with the tail-count rule, the device weight pack against the host packer
# on the hardware, and the training step with eight-wave tall products
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_backward.py -m gpu -q --no-header -p no:cacheprovider -s -k "margin_protocol or device_weight_pack or bwd_gemm or elementwise or dense_backward_on or full_backward" > $OUT/k_pytest.log 2>&1; echo "[pytest rc=$?]"
tail -3 $OUT/k_pytest.log | cut -c1-300; grep "^FAILED\|^ERROR\|^E  " $OUT/k_pytest.log | cut -c1-300 | head -12
timeout 300 python bench_train.py --steps 4 --warmup 2 > $OUT/k_train.json 2> $OUT/k_train.err; echo "[train rc=$?]"; cut -c1-700 $OUT/k_train.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/k_prof -o trace -- python $GRAFT_REPO_ROOT/bench_train.py --steps 3 --warmup 1 > $OUT/k_prof.log 2>&1; echo "[rocprof rc=$?]"
DB=$(find $OUT/k_prof -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 45 > $OUT/k_prof_stats.txt; head -16 $OUT/k_prof_stats.txt | cut -c1-170
find $OUT/k_prof -name "*.db" -size +20M -delete
