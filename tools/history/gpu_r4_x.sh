#!/bin/bash
# round 4, call X: the default bench line and the whole GPU suite at HEAD
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
bash tools/gpu_r4_v.sh
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -s > $OUT/x_pytest.log 2>&1; echo "[pytest rc=$?]"
tail -4 $OUT/x_pytest.log | cut -c1-300; grep "^FAILED\|^ERROR" $OUT/x_pytest.log | cut -c1-300 | head -20
