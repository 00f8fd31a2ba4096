#!/bin/bash
# round 4, call J: the new parity-hardening tests on the MI355X + the full-size frame checks in the headline configuration
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_backward.py -x -q -s -k "full_size_backward" 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_producers.py -x -q -s -k "golden" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "full_size_frame_properties and enc16" 2>&1 | tail -15
} > gpurun_out/r4_j.log 2>&1
tail -60 gpurun_out/r4_j.log
