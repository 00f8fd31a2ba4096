#!/bin/bash
# round 6, call L: the sixteen-channel-per-lane gather (two lanes per sample; the default now) against the eight-channel one (SHERF_EXPERIMENT bit 10) and round 5's
# loop (bits 10 + 9): whole-frame A/B in one process, bits + timeline; then quick parity tests through the new kernel
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python tools/frame_ab.py --config cfg2_dense_ri --arms 0,0,0 --exps 0,1024,1536 --names h16,h8_prefetch,h8_round5 --timeline --rounds 4 > $OUT/r6l_frame_ab.log 2>&1
echo "[frame_ab rc=$?]"; grep "^\[\|configuration" $OUT/r6l_frame_ab.log | cut -c1-400
timeout 900 python tools/frame_ab.py --config cfg2_ri --arms 0,0 --exps 0,1024 --names h16,h8_prefetch --rounds 3 > $OUT/r6l_frame_ab_cfg2.log 2>&1
echo "[frame_ab cfg2_ri rc=$?]"; grep "^\[arm\|^\[bits" $OUT/r6l_frame_ab_cfg2.log | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu --no-header -p no:cacheprovider -x -k "gathered_tokens or end_to_end or margin_protocol or ragged or no_valid" > $OUT/r6l_pytest.log 2>&1
echo "[pytest rc=$?]"; tail -3 $OUT/r6l_pytest.log | cut -c1-300
