#!/bin/bash
# round 4, call O: the whole GPU suite at HEAD (token-side workspace sized per frame, near lists, host-side descriptor cache)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x -s > $OUT/o_pytest.log 2>&1; echo "[pytest rc=$?]"
tail -5 $OUT/o_pytest.log | cut -c1-300; grep "^FAILED\|^ERROR\|^E  " $OUT/o_pytest.log | cut -c1-300 | head -20
