#!/bin/bash
# round 6, call AA: the flag watch reading the counters every eighth frame (every frame's flags fold into sticky words on the device: sherf_frame.sticky): tools/tail_probe.py
# (forward() against the natively looped frame), the GPU tests around the watch / token workspace / calibration, then the default bench line
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python tools/tail_probe.py > $OUT/r6aa_tail_probe.txt 2>&1; echo "[probe rc=$?]"; grep "^\[arm" $OUT/r6aa_tail_probe.txt
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_glue.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
timeout 900 python bench.py --no-secondary --no-train --no-cpu-baseline --no-pmc > $OUT/r6aa_bench.json 2> $OUT/r6aa_bench.err; echo "[bench rc=$?]"; python -c "
import json; d=json.loads(open('$OUT/r6aa_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('parity_ok'), d['config']['mlp_form'], d['config']['mlp_form_auto'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['frame_timeline_ms'])"
