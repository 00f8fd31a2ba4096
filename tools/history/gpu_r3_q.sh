#!/bin/bash
# round 3, call Q: the backward's GPU tests and the training step with the fp16-split forward recompute
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 100 python -m pytest tests/test_gpu_backward.py -m gpu -q --no-header -p no:cacheprovider -x > $OUT/q_pytest.log 2>&1; echo "[pytest rc=$?]"; tail -3 $OUT/q_pytest.log | cut -c1-300; grep "^FAILED\|^ERROR\|^E  " $OUT/q_pytest.log | cut -c1-300 | head
timeout 60 python bench_train.py --steps 4 --warmup 2 > $OUT/q_train.json 2> $OUT/q_train.err; echo "[train rc=$?]"; cut -c1-600 $OUT/q_train.json
