#!/bin/bash
# round 4, call Q: the token-workspace policy on the real runtime + the final evidence run
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -x -s -k "token_workspace or auto_precision or deterministic" > $OUT/q_pytest.log 2>&1; echo "[pytest rc=$?]"; tail -4 $OUT/q_pytest.log | cut -c1-300; grep "^FAILED\|^ERROR\|^E  " $OUT/q_pytest.log | cut -c1-300 | head
bash tools/gpu_r4_final.sh
