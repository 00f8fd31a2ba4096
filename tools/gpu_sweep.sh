#!/bin/bash
# scheduling sweep for the frame driver (HIP-event timelines from bench.py, no profiler)
export SHERF_MLP_SHAPE=${SHERF_MLP_SHAPE:-8x1}   # A/B runs pin the MLP shape (bench.py would otherwise autotune it)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
run() { # label, bn mode, env...
  local label=$1; local bn=$2; shift; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --bn-mode $bn 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['frame_timeline_ms']; print('%-18s' % '$label', 'ms/step %.3f' % d['ms_per_step'], 'mlp %.3f' % d['roofline']['kernel_ms'], ' '.join('%s=%.3f' % (k[:12], v) for k, v in t.items()))"
}
run eval_conc eval X=1
run eval_conc_prio eval SHERF_DEBUG=128
run eval_noaux eval SHERF_AUX_STREAM=0
run eval_serial eval SHERF_MAIN_AFTER_LAYER=13
run train_conc train X=1
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -x --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $OUT/pytest.log
