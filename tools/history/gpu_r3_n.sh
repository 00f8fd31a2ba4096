#!/bin/bash
# round 3, call N: the whole GPU suite and the training step at HEAD (after the final run's follow-ups)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python bench_train.py --steps 4 --warmup 2 > $OUT/n_train.json 2> $OUT/n_train.err; echo "[train rc=$?]"; cut -c1-700 $OUT/n_train.json
timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -s > $OUT/n_pytest.log 2>&1; echo "[pytest rc=$?]"; tail -4 $OUT/n_pytest.log | cut -c1-300; grep "^FAILED\|^ERROR" $OUT/n_pytest.log | cut -c1-300 | head
