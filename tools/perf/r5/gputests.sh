#!/bin/bash
# the GPU test suite as the driver runs it, output kept under gpurun_out/
set -o pipefail
mkdir -p gpurun_out/r5_tests
export MEMB_SYNTH_DEVICE=0
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > gpurun_out/r5_tests/gpu.log 2>&1
code=$?
tail -15 gpurun_out/r5_tests/gpu.log
exit $code
