#!/bin/bash
# round 5, call I: TriPlaneGenerator.forward with its producers replayed as hipGraphs (--graph-producers) against the eager producers
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for flags in "--eager-producers" ""; do   # (as run: "" and "--graph-producers" -- replay became the default afterwards, the flag now selects the eager producers)
  timeout 400 python bench_generator.py --steps 30 --warmup 8 $flags > $OUT/r5i_gen.json 2> $OUT/r5i_gen.err; echo "[bench_generator '$flags' rc=$?]"
  python - <<'PY'
import json
try:
    d = json.loads([l for l in open('/root/repo/gpurun_out/r5i_gen.json') if l.startswith('{')][-1])
    print('   recomputed', round(d['recomputed_every_frame']['ms_per_forward'], 3), d['recomputed_every_frame']['stages_ms'])
    print('   cached    ', round(d['use_cached_backbone']['ms_per_forward'], 3), d['use_cached_backbone']['stages_ms'])
    print('   graphed', d.get('graphed'), 'output', d['output'])
except Exception as ex:
    print('   no line', ex)
PY
  tail -2 $OUT/r5i_gen.err | cut -c1-300
  cp $OUT/r5i_gen.json "$OUT/r5i_gen_${flags:-eager}.json"
done
