#!/bin/bash
# kernel-variant comparison: alternative builds of the library (same ABI) selected with SHERF_HIP_LIB
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for V in "" _pf2 _pf3; do
  export SHERF_HIP_LIB=$GRAFT_REPO_ROOT/sherf_amd/libsherf_hip$V.so
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | grep '"metric"\|rror' | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['frame_timeline_ms']; print('variant[$V]', 'ms/step %.3f' % d['ms_per_step'], 'mlp %.4f' % d['roofline']['kernel_ms'], 'frac %.4f' % d['roofline']['frac'])"
done
