#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -x --no-header -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 $OUT/pytest.log
cd /tmp
for F in 0; do
  rm -rf $OUT/abl_$F; mkdir -p $OUT/abl_$F
  SHERF_DEBUG=$F timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/abl_$F -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/abl_$F.log 2>&1
  if [ "$F" = "0" ]; then python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $OUT/abl_0/t_results.db > $OUT/timeline.txt; python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $OUT/abl_0/t_results.db 13 45 > $OUT/kernel_stats.txt; cut -c1-150 $OUT/kernel_stats.txt | head -24; grep '"metric"' $OUT/abl_0.log | cut -c1-200; fi
  python - $OUT/abl_$F/t_results.db $F <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), avg(end-start) from kernels group by name order by sum(end-start) desc").fetchall()
pick = [r for r in rows if any(k in r[0] for k in ('sample_nn', 'gather_tokens', 'nerf_mlp', 'build_cells'))]
print('flags', sys.argv[2], ' | '.join(f"{r[0].split('::')[-1].split('(')[0][:22]} {r[2]/1e3:.0f}us" for r in pick))
PY
  rm -rf $OUT/abl_$F
done
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep '"metric"' | tee $OUT/bench.log | cut -c1-260
