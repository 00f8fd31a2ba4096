#!/bin/bash
# MLP ring depth + LDS cell build: timelines, then the GPU test-suite
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() { # label, bn mode, env...
  local label=$1; local bn=$2; shift; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --bn-mode $bn 2>&1 | grep '"metric"\|rror' | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['frame_timeline_ms']; print('%-18s' % '$label', 'ms/step %.3f' % d['ms_per_step'], 'mlp %.3f' % d['roofline']['kernel_ms'], ' '.join('%s=%.3f' % (k[:12], v) for k, v in t.items()))"
}
run eval_8x1 eval X=1
run eval_8x1s4 eval SHERF_MLP_SHAPE=8x1s4
run eval_8x1s5 eval SHERF_MLP_SHAPE=8x1s5
run train_8x1 train X=1
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -x --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $OUT/pytest.log
