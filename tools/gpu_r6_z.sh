#!/bin/bash
# round 6, call Z: mlp_form='auto' timing whole frames (instead of the kernel alone, back to back): the tuner's report beside the pinned forms of the same process, the GPU
# tests that touch the forms, then the default bench line
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python tools/frame_ab.py --config cfg2_dense_ri --arms 0,0,0,0 --names auto,pipelined,two_tiles,one --opts "mlp_form='auto';mlp_form='pipelined';mlp_form='two_tiles';mlp_form='one'" --timeline --rounds 5 > $OUT/r6z_frame_ab.log 2>&1
echo "[ab rc=$?]"; grep "^\[timeline\|^\[arm\|^\[bits\|^\[mlp_form" $OUT/r6z_frame_ab.log | cut -c1-330; tail -3 $OUT/r6z_frame_ab.log | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_glue.py -m gpu -q -x -p no:cacheprovider -k "forms or round_6 or two_launch" 2>&1 | tail -3
timeout 600 python bench.py --no-secondary --no-train --no-cpu-baseline --no-torch-gpu-baseline --no-pmc > $OUT/r6z_bench.json 2> $OUT/r6z_bench.err; echo "[bench rc=$?]"; python -c "
import json; d=json.loads(open('$OUT/r6z_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['parity_ok'], d['config']['mlp_form'], d['config']['mlp_form_auto'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
