#!/bin/bash
# round 5, call G: TriPlaneGenerator.forward at full size; full-size backward against the float64 truth (both framings); new GPU tests (ResNet-18 vs an
# independent formulation, density noise); power / clock telemetry under the network kernel; counters of the network and the gather on the dense frame
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python bench_generator.py --steps 20 --warmup 5 > $OUT/r5g_bench_generator.json 2> $OUT/r5g_bench_generator.err; echo "[bench_generator rc=$?]"; tail -c 2600 $OUT/r5g_bench_generator.json; tail -3 $OUT/r5g_bench_generator.err
timeout 900 python -m pytest tests/test_gpu_backward.py -q -x -m gpu -s -k "full_size_backward" > $OUT/r5g_pytest_backward_truth64.txt 2>&1; echo "[pytest backward rc=$?]"; grep "gradients, loss\|^   renderer\|^   input\|^   decoder\|worst outside\|passed\|failed\|Error" $OUT/r5g_pytest_backward_truth64.txt | head -40
timeout 600 python -m pytest tests/test_gpu_producers.py -q -x -m gpu -k "resnet18 or density_noise" > $OUT/r5g_pytest_resnet_noise.txt 2>&1; echo "[pytest resnet/noise rc=$?]"; tail -3 $OUT/r5g_pytest_resnet_noise.txt
timeout 300 python tools/power_probe.py -- python tools/mlp_ab.py --config cfg2_dense_ri --forms one,pp --rounds 8 --out $OUT/r5g_mlp_ab_power.json > $OUT/r5g_power_probe.txt 2>&1; echo "[power probe rc=$?]"; grep "power_probe\|^\[arm" $OUT/r5g_power_probe.txt | cut -c1-200
timeout 200 python tools/power_probe.py -- python tools/mlp_ab.py --config cfg2_dense_ri --forms one,pp --rounds 8 --zero all --out $OUT/r5g_mlp_ab_power_zero.json > $OUT/r5g_power_probe_zero.txt 2>&1; echo "[power probe zero rc=$?]"; grep "power_probe\|^\[arm" $OUT/r5g_power_probe_zero.txt | cut -c1-200
cd /tmp
C="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --precision f16 --no-cpu-baseline --no-torch-gpu-baseline --no-secondary --no-pmc --no-train"
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
  TAG=$(echo $SET | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $SET -d $OUT/r5g_pmc_$TAG -o pmc -- $C > $OUT/r5g_pmc_$TAG.log 2>&1; echo "[pmc $TAG rc=$?]"
  DB=$(find $OUT/r5g_pmc_$TAG -name "*.db" | head -1)
  [ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/pmc_query.py $DB gather_tokens nerf_mlp 2>&1 | cut -c1-120 | grep -v "^# pmc" | tee $OUT/r5g_pmc_$TAG.txt | head -30
  rm -rf $OUT/r5g_pmc_$TAG
done
