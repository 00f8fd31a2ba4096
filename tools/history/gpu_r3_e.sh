#!/bin/bash
# round 3, call E: eight-channel fp16 gather + XCD-banded tile order (A/B through debug bits), single-product encoder diagnostic
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python tools/enc_sp_diag.py tiny_ri > $OUT/e_enc_diag.log 2>&1; echo "[enc diag rc=$?]"; grep -v amdgpu.ids $OUT/e_enc_diag.log | cut -c1-220 | tail -22
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -s -k "auto or gathered or (full_size and ri) or end_to_end" > $OUT/e_pytest.log 2>&1; echo "[pytest rc=$?]"
tail -3 $OUT/e_pytest.log | cut -c1-300; grep "^FAILED\|^ERROR\|auto ->\|clean " $OUT/e_pytest.log | cut -c1-300 | head -12
Q="--steps 40 --warmup 10 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc"
run() { env $1 timeout 300 python bench.py $Q $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d.get('frame_timeline_ms'); print('$1 $2', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,1), 'Mrays/s mlp', round(d['roofline']['kernel_ms'],4), d['config'].get('mlp_precision'), d['config'].get('table_precision'), d['config'].get('encoder_precision'), 'gather', round(t['gather_done']-max(t['encoder_done'],t['rays_at_encoder_join']),4), t)"; }
run X=0 "--precision f16"
run SHERF_DEBUG=1024 "--precision f16"
run SHERF_DEBUG=2048 "--precision f16"
run SHERF_DEBUG=3072 "--precision f16"
run X=0 "--precision f16 --table-precision f32"
run SHERF_DEBUG=1024 "--precision f16 --table-precision f32"
run X=0 ""
run X=0 "--config cfg2_dense_ri"
run X=0 "--config cfg2 --precision f16x3"
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 6 --precision f16 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/e_prof -o trace -- $B > $OUT/e_prof.log 2>&1; echo "[rocprof rc=$?]"
DB=$(find $OUT/e_prof -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 45 > $OUT/e_prof_stats.txt; head -12 $OUT/e_prof_stats.txt | cut -c1-150
find $OUT/e_prof -name "*.db" -size +20M -delete
