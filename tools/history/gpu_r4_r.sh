#!/bin/bash
# round 4, call R: the token-workspace policy on the real runtime
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -x -s -k "token_workspace or mask_and_nearest or warp_matches" > $OUT/r_pytest.log 2>&1; echo "[pytest rc=$?]"; tail -4 $OUT/r_pytest.log | cut -c1-300; grep "^FAILED\|^ERROR\|^E  " $OUT/r_pytest.log | cut -c1-300 | head
