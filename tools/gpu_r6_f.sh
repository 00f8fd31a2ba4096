#!/bin/bash
# round 6, call F: capture probes inside the product process (tools/graph_probe.py)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -c "
import torch, ctypes, os
print('torch hip', torch.version.hip)
import subprocess
print(subprocess.run('grep -a libamdhip /proc/self/maps | head -0; ldd sherf_amd/libsherf_hip.so | grep -i hip', shell=True, capture_output=True, text=True).stdout)
"
timeout 900 python tools/graph_probe.py 2>&1 | tee $OUT/r6f_graph_probe.txt | tail -60
