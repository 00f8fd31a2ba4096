#!/bin/bash
# scheduling sweep for the two-stream frame driver
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['frame_timeline_ms']; print('%-18s' % '$label', 'ms/step %.3f' % d['ms_per_step'], 'mlp %.3f' % d['roofline']['kernel_ms'], ' '.join('%s=%.3f' % (k[:12], v) for k, v in t.items()))"
}
run concurrent X=1
run concurrent_split SHERF_GATHER_SPLIT=1
run after1 SHERF_MAIN_AFTER_LAYER=1
run after4 SHERF_MAIN_AFTER_LAYER=4
run after4_split SHERF_GATHER_SPLIT=1 SHERF_MAIN_AFTER_LAYER=4
run serial SHERF_MAIN_AFTER_LAYER=13
run concurrent_prio SHERF_DEBUG=128
run concurrent_prio_split SHERF_DEBUG=128 SHERF_GATHER_SPLIT=1
