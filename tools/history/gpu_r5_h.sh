#!/bin/bash
# round 5, call H: power / clock telemetry under a SUSTAINED network-kernel load (real and zeroed data); kernel trace of TriPlaneGenerator.forward; MIOpen knobs
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for z in "" all; do
  L=$OUT/r5h_sustain_${z:-real}.log
  POWER_PROBE_LOG=$L timeout 200 python tools/power_probe.py -- bash -c "python tools/mlp_ab.py --config cfg2_dense_ri --forms pp --rounds 2 --sustain 4 ${z:+--zero $z} --out $OUT/r5h_ab_${z:-real}.json > $L 2>&1" > $OUT/r5h_power_${z:-real}.txt 2>&1
  echo "[sustain ${z:-real} rc=$?]"; grep "^\[arm\|^\[sustain" $L | cut -c1-160; grep power_probe $OUT/r5h_power_${z:-real}.txt | cut -c1-200
done
for flags in "" "--miopen-benchmark" "--channels-last" "--miopen-benchmark --channels-last"; do
  timeout 400 python bench_generator.py --steps 20 --warmup 8 $flags > $OUT/r5h_gen.json 2> $OUT/r5h_gen.err; echo "[bench_generator '$flags' rc=$?]"
  python - <<'PY'
import json
try:
    d = json.loads([l for l in open('/root/repo/gpurun_out/r5h_gen.json') if l.startswith('{')][-1])
    print('   recomputed', round(d['recomputed_every_frame']['ms_per_forward'], 3), d['recomputed_every_frame']['stages_ms'])
    print('   cached    ', round(d['use_cached_backbone']['ms_per_forward'], 3), d['use_cached_backbone']['stages_ms'])
except Exception as ex:
    print('   no line', ex)
PY
  tail -2 $OUT/r5h_gen.err | cut -c1-200
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/r5h_gen_trace -o gen -- python $GRAFT_REPO_ROOT/bench_generator.py --steps 10 --warmup 4 > $OUT/r5h_gen_trace.log 2>&1; echo "[gen trace rc=$?]"
F=$(find $OUT/r5h_gen_trace -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -30 $F | cut -c1-200 | tee $OUT/r5h_gen_kernel_stats_head.txt
DB=$(find $OUT/r5h_gen_trace -name "*.db" | head -1); [ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB 2>/dev/null | head -45 | cut -c1-200 | tee $OUT/r5h_gen_kernel_stats.txt
rm -rf $OUT/r5h_gen_trace
