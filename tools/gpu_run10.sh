#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() { # label, bn mode, env...
  local label=$1; local bn=$2; shift; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --bn-mode $bn 2>&1 | grep '"metric"\|rror' | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['frame_timeline_ms']; print('%-18s' % '$label', 'ms/step %.3f' % d['ms_per_step'], 'mlp %.3f' % d['roofline']['kernel_ms'], ' '.join('%s=%.3f' % (k[:12], v) for k, v in t.items()))"
}
run train train X=1
run train2 train X=1
run eval eval X=1
timeout 600 python -m pytest tests -m gpu -q --timeout=600 -x --no-header -p no:cacheprovider -k "voxel or end_to_end or eval_mode or deterministic or tokens or generator or synthesis" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 $OUT/pytest.log
