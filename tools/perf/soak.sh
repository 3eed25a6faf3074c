#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/soak
export MEMB_SYNTH_DEVICE=0
SOAK_SECONDS=${SOAK_SECONDS:-420} SOAK_SEED=${SOAK_SEED:-55} timeout -k 10 1100 python tools/perf/soak.py > gpurun_out/soak/soak_$SOAK_SEED.txt 2>&1
code=$?
tail -5 gpurun_out/soak/soak_$SOAK_SEED.txt
exit $code
