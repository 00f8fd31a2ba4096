#!/bin/bash
# First GPU run of the next session: (1) the forward suite as a sanity check, (2) every staged backward test in its OWN process
# (a faulting kernel must not poison the others), (3) the prepared kernel variants A/B against the default build.
#   bash tools/build_variants.sh && gpurun --timeout 1500 -- 'bash tools/gpu_next.sh'
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --timeout=600 --no-header -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; tail -3 $OUT/pytest_gpu.log
: > $OUT/pytest_experimental.log
for T in $(python -m pytest tests/test_gpu_backward.py -m gpu_experimental --collect-only -q -p no:cacheprovider 2>/dev/null | grep "::"); do
  timeout 300 python -m pytest "$T" -m gpu_experimental -q --no-header -p no:cacheprovider -x > $OUT/one.log 2>&1; rc=$?
  echo "[$rc] $T" | tee -a $OUT/pytest_experimental.log
  if [ $rc -ne 0 ]; then grep -E "Error|error|assert|rel\(|failed" $OUT/one.log | head -12 >> $OUT/pytest_experimental.log; fi
done
V=""
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu_experimental -q -s --no-header -p no:cacheprovider 2>&1 | tail -6 | tee -a $OUT/pytest_experimental.log
timeout 300 python -m pytest tests/test_gpu_tune.py -m gpu_experimental -q -s --no-header -p no:cacheprovider 2>&1 | tail -40 | tee -a $OUT/pytest_experimental.log
for t in il8 erf erf_il8 gbl prio prio_il8; do [ -f sherf_amd/libsherf_hip_$t.so ] && V="$V $t"; done
bash tools/gpu_variants.sh $V 2>&1 | tee $OUT/variants.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '"metric"' | tee $OUT/bench_auto.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('auto      ', d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['mlp_shape'], d['config']['gather']); [print('   ', k, v) for k, v in d.get('mlp_tune', {}).get('shapes', {}).items()]; print('    gather', d.get('mlp_tune', {}).get('gather'))"
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --mlp-shape 8x1 2>/dev/null | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default   ', d['ms_per_step'], d['roofline']['kernel_ms'])"
SHERF_MLP_SHAPE=8x1split python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('8x1split  ', d['ms_per_step'], d['roofline']['kernel_ms'], '(kernel_ms covers both launches)')"
SHERF_MLP_SHAPE=8x1split2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('8x1split2 ', d['ms_per_step'], d['roofline']['kernel_ms'], '(decoder walks two output tiles per step)')"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --torch-gpu-baseline 2>/dev/null | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stock-ops oracle on the GPU:', d.get('torch_gpu_baseline'))"
SHERF_MLP_SHAPE=8x1persist python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('8x1persist', d['ms_per_step'], d['roofline']['kernel_ms'], '(persistent workgroups, ring streams across tile groups)')"
