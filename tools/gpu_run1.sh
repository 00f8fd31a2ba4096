#!/bin/bash
# first GPU session: diagnostics, parity tests, short bench, kernel-trace profile
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/gpu_diag.py tiny > gpurun_out/diag.log 2>&1; echo "diag rc=$?"
tail -40 gpurun_out/diag.log
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -x --no-header -p no:cacheprovider > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/pytest.log
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
tail -5 gpurun_out/bench.log
