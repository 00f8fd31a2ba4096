#!/bin/bash
# round 6, call B: frames as hipGraphs -- the new GPU tests (graph replay, checked producer graphs, feature-branch switches, encodings-in-gather bits),
# whole-frame A/B graphs on / off (bits + timeline + host time), then the default bench line
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_producers.py tests/test_gpu_glue.py -q -m gpu --no-header -p no:cacheprovider \
  -k "hipgraphs or graphed_producer or feature_branch or encodings_in_the_gather or deterministic or token_workspace or without_transformer" > $OUT/r6b_pytest.log 2>&1
echo "[pytest rc=$?]"; tail -8 $OUT/r6b_pytest.log | cut -c1-300
timeout 600 python tools/frame_ab.py --config cfg2_dense_ri --arms 0,0 --names eager,graph --opts "frame_graph=False;frame_graph=True" --timeline --rounds 4 > $OUT/r6b_frame_ab.log 2>&1
echo "[frame_ab rc=$?]"; grep "^\[\|configuration" $OUT/r6b_frame_ab.log | cut -c1-400
timeout 600 python tools/frame_ab.py --config cfg2_ri --arms 0,0 --names eager,graph --opts "frame_graph=False;frame_graph=True" --rounds 3 > $OUT/r6b_frame_ab_cfg2.log 2>&1
echo "[frame_ab cfg2_ri rc=$?]"; grep "^\[\|configuration" $OUT/r6b_frame_ab_cfg2.log | cut -c1-400
timeout 900 python bench.py > $OUT/r6b_bench.json 2> $OUT/r6b_bench.err; echo "[bench rc=$?]"; tail -3 $OUT/r6b_bench.err | cut -c1-300; python -c "
import json; d=json.loads(open('$OUT/r6b_bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in list(d)[:12]})
print('roofline', {k: d['roofline'].get(k) for k in ('kernel','frac','kernel_ms','traffic')}); print('parity_ok', d.get('parity_ok'))
print('graphs', d['config'].get('frame_graphs')); print('timeline', d.get('frame_timeline_ms'))
print('secondary keys', list((d.get('secondary') or {}).keys()))"
