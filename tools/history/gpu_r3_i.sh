#!/bin/bash
# round 3, call I: where the binned tap scatter spends its time (ablations), then the step
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python tools/scatter_bench.py cfg2 > $OUT/i_scatter.txt 2>&1; echo "[scatter rc=$?]"; cat $OUT/i_scatter.txt | cut -c1-200
timeout 300 python -m pytest tests/test_gpu_backward.py -m gpu -q --no-header -p no:cacheprovider -k "gather_backward or full_backward" > $OUT/i_pytest.log 2>&1; echo "[pytest rc=$?]"; tail -2 $OUT/i_pytest.log | cut -c1-200
timeout 300 python bench_train.py --steps 4 --warmup 2 > $OUT/i_train.json 2> $OUT/i_train.err; echo "[train rc=$?]"; cut -c1-900 $OUT/i_train.json
