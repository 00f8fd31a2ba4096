#!/bin/bash
# round 3, call F: sparse convolutions over eight waves (debug bit 12 = four), candidate search with two candidates in flight per group
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider -s -k "sparse_voxel or mask_and or (full_size and cfg2_ri) or end_to_end or ragged or deterministic" > $OUT/f_pytest.log 2>&1; echo "[pytest rc=$?]"
tail -3 $OUT/f_pytest.log | cut -c1-300; grep "^FAILED\|^ERROR\|clean " $OUT/f_pytest.log | cut -c1-300 | head -12
Q="--steps 40 --warmup 10 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc"
run() { env $1 timeout 300 python bench.py $Q $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d.get('frame_timeline_ms'); print('$1 $2', round(d['ms_per_step'],4), 'ms', round(d['value']/1e6,1), 'Mrays/s mlp', round(d['roofline']['kernel_ms'],4), d['config'].get('mlp_precision'), d['config'].get('table_precision'), d['config'].get('encoder_precision'), 'gather', round(t['gather_done']-max(t['encoder_done'],t['rays_at_encoder_join']),4), t)"; }
run X=0 "--precision f16"
run SHERF_DEBUG=4096 "--precision f16"
run X=0 "--precision f16"
run SHERF_DEBUG=4096 "--precision f16"
run X=0 "--config cfg2_dense_ri --precision f16"
run X=0 "--config cfg2 --precision f16x3"
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 6 --precision f16 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/f_prof -o trace -- $B > $OUT/f_prof.log 2>&1; echo "[rocprof rc=$?]"
DB=$(find $OUT/f_prof -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 0 45 > $OUT/f_prof_stats.txt; python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB > $OUT/f_prof_timeline.txt 2>&1; head -22 $OUT/f_prof_stats.txt | cut -c1-150
find $OUT/f_prof -name "*.db" -size +20M -delete
