#!/bin/bash
# round 6, call A: the encodings-in-the-gather path on the MI355X -- quick parity tests, whole-frame A/B (bits + timeline), the network kernel alone
# (pp vs pe, the MFMA-only / no-encodings / no-transformer ablation builds), rocm-smi power + clocks under the new kernel, then the default bench line
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_producers.py -q -m gpu --no-header -p no:cacheprovider -x \
  -k "native_library or gathered_tokens or per_sample_sigma or end_to_end or auto_precision or graphed_producer or margin_protocol" > $OUT/r6a_pytest.log 2>&1
echo "[pytest rc=$?]"; tail -5 $OUT/r6a_pytest.log | cut -c1-300
timeout 600 python tools/frame_ab.py --config cfg2_dense_ri --arms 0,0 --names pe_off,pe_on --opts "pe_in_gather=False;pe_in_gather=True" --timeline --rounds 4 > $OUT/r6a_frame_ab.log 2>&1
echo "[frame_ab rc=$?]"; grep "^\[\|configuration" $OUT/r6a_frame_ab.log | cut -c1-400
timeout 600 python tools/mlp_ab.py --config cfg2_dense_ri --precision f16 --forms pp,pe --rounds 3 --stress 30 --out $OUT/r6a_mlp_ab.json > $OUT/r6a_mlp_ab.log 2>&1
echo "[mlp_ab rc=$?]"; grep "^\[" $OUT/r6a_mlp_ab.log | cut -c1-200
# power / clock telemetry under the sustained new kernel
L=$OUT/r6a_sustain.log
python tools/mlp_ab.py --only product --config cfg2_dense_ri --precision f16 --forms pp,pe --rounds 1 --sustain 7 --out $OUT/r6a_sustain.json > $L 2>&1 &
PID=$!
for i in $(seq 1 240); do grep -q "^\[sustain\] start" $L 2>/dev/null && break; sleep 0.5; done
sleep 1.0
for k in 1 2 3; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | head -10 | tee -a $OUT/r6a_smi.txt
  sleep 0.7
done
wait $PID
grep "^\[sustain" $L | cut -c1-200
timeout 900 python bench.py > $OUT/r6a_bench.json 2> $OUT/r6a_bench.err; echo "[bench rc=$?]"; python -c "
import json; d=json.loads(open('$OUT/r6a_bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in list(d)[:12]})
print('roofline', {k: d['roofline'].get(k) for k in ('kernel','frac','kernel_ms','traffic')}); print('parity_ok', d.get('parity_ok'))
print('timeline', d.get('frame_timeline_ms'))"
