#!/bin/bash
# round 6, call I: the fp16 path of the tri-plane generator (device test vs the reference golden, bench_generator variant), the opt-in frame graphs' test,
# the checked producer graphs, then the default bench line at this commit (incl. `train` with counted scatter traffic and `train_dense`)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_producers.py -q -m gpu --no-header -p no:cacheprovider -s \
  -k "hipgraphs or graphed_producer or fp16_path or full_size_stylegan2" > $OUT/r6i_pytest.log 2>&1
echo "[pytest rc=$?]"; grep "full-size backbone\|passed\|failed\|Error" $OUT/r6i_pytest.log | cut -c1-250 | tail -8
timeout 600 python bench_generator.py > $OUT/r6i_bench_generator.json 2> $OUT/r6i_bench_generator.err; echo "[bench_generator rc=$?]"; python -c "
import json; d=json.loads(open('$OUT/r6i_bench_generator.json').read().strip().splitlines()[-1])
for k in ('recomputed_every_frame','recomputed_every_frame_backbone_fp16','use_cached_backbone','recomputed_every_frame_eager_producers'):
    print(k, json.dumps(d.get(k))[:600])
print('graphed', d.get('graphed'))"
timeout 1500 python bench.py > $OUT/r6i_bench.json 2> $OUT/r6i_bench.err; echo "[bench rc=$?]"; tail -2 $OUT/r6i_bench.err | cut -c1-300; python -c "
import json; d=json.loads(open('$OUT/r6i_bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in list(d)[:12]})
print('roofline', {k: d['roofline'].get(k) for k in ('kernel','frac','kernel_ms','traffic')}); print('parity_ok', d.get('parity_ok'))
print('graphs', d['config'].get('frame_graphs'), 'pe', d['config'].get('pe_in_gather')); print('timeline', d.get('frame_timeline_ms'))
print('train', json.dumps(d.get('train'))[:900]); print('train_dense', json.dumps(d.get('train_dense'))[:500])
print('generator', json.dumps((d.get('secondary') or {}).get('generator_forward'))[:700])"
