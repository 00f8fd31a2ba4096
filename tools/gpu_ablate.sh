#!/bin/bash
# ablation timing of the sampler / gather (SHERF_DEBUG bits) + one PMC pass; profiling only
export SHERF_MLP_SHAPE=${SHERF_MLP_SHAPE:-8x1}   # A/B runs pin the MLP shape (bench.py would otherwise autotune it)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=600 -x --no-header -p no:cacheprovider -k "deterministic or voxel or end_to_end" -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "ray-independence|passed|failed|PSNR" $OUT/pytest.log | tail -8
cd /tmp
for F in 0 1 2 4 8 16 28; do
  rm -rf $OUT/abl_$F; mkdir -p $OUT/abl_$F
  SHERF_DEBUG=$F timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/abl_$F -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/abl_$F.log 2>&1
  python - $OUT/abl_$F/t_results.db $F <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), avg(end-start) from kernels group by name order by sum(end-start) desc").fetchall()
pick = [r for r in rows if any(k in r[0] for k in ('sample_nn', 'gather_tokens', 'nerf_mlp', 'warp_geom', 'compact'))]
print('flags', sys.argv[2], ' | '.join(f"{r[0].split('::')[-1].split('(')[0][:22]} {r[2]/1e3:.0f}us" for r in pick))
PY
done
rm -rf $OUT/abl_*/
# PMC pass (own run, kernel-trace only alongside)
rm -rf $OUT/pmc; mkdir -p $OUT/pmc
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/pmc -o p1 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc1.log 2>&1; echo "pmc1 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/pmc -o p2 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc2.log 2>&1; echo "pmc2 rc=$?"
ls -la $OUT/pmc | head; tail -3 $OUT/pmc1.log | cut -c1-300
