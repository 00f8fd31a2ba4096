#!/bin/bash
# round 5, call M: rocm-smi / amd-smi clock + power readings while nerf_mlp3_kernel runs back to back for 6 s (real data, then zeroed data)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
which rocm-smi amd-smi 2>&1 | head -3
for z in "" all; do
  L=$OUT/r5m_sustain_${z:-real}.log
  python tools/mlp_ab.py --config cfg2_dense_ri --forms pp --rounds 2 --sustain 7 ${z:+--zero $z} --out $OUT/r5m_ab_${z:-real}.json > $L 2>&1 &
  PID=$!
  # wait for the sustained window to open
  for i in $(seq 1 120); do grep -q "^\[sustain\] start" $L 2>/dev/null && break; sleep 0.5; done
  sleep 1.0
  echo "== data: ${z:-real}" | tee -a $OUT/r5m_smi.txt
  for k in 1 2 3; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|fclk\|mclk" | head -8 | tee -a $OUT/r5m_smi.txt
    sleep 0.7
  done
  (amd-smi metric -g 0 --clock --power 2>/dev/null | head -60) | tee -a $OUT/r5m_amdsmi_${z:-real}.txt | grep -i "clk\|power\|GFX_0\|MIN\|MAX\|CUR" | head -30
  wait $PID
  grep "^\[arm\|^\[sustain" $L | cut -c1-170
done
