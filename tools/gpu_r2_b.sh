#!/bin/bash
# round 2, call B: in-kernel timeline of the MLP + -D variants + ablations; re-run of the tests fixed after call A
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
L=sherf_amd
timeout 600 python tools/mlp_trace.py --shapes 4x1,8x1,4x1phase --time-libs trace=$L/libsherf_hip_trace.so,erf=$L/libsherf_hip_erf.so,noslp=$L/libsherf_hip_noslp.so,splitk=$L/libsherf_hip_splitk.so > $OUT/mlp_trace.log 2>&1; echo "[trace rc=$?]"; grep "^\[" $OUT/mlp_trace.log; grep "^ " $OUT/mlp_trace.log | head -40
for T in "tests/test_gpu_backward.py::test_composite_backward_kernel" "tests/test_gpu_backward.py::test_full_backward_against_reference_gradients" "tests/test_gpu_ops.py::test_bias_act_layouts"; do
  timeout 300 python -m pytest "$T" -m gpu_experimental -q --no-header -p no:cacheprovider -x > $OUT/one.log 2>&1; rc=$?
  echo "[$rc] $T"; if [ $rc -ne 0 ]; then tail -30 $OUT/one.log; fi
done
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -k "dataset_rays or units" > $OUT/one.log 2>&1; echo "[$?] rays"; tail -15 $OUT/one.log
