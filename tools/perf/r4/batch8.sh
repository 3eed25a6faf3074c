#!/bin/bash
# Round 4, batch 8: uniform kernels with the reciprocal + two-FMA division (build/measure = the tree) against the tree
# before it (build/prev, hipcc's correctly rounded division): alternating processes on one box; then the new GPU test.
set -o pipefail
out=gpurun_out/r4_batch8
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
for round in 1 2 3; do
    for root in prev measure; do
        MEMB_PACKAGE_ROOT=build/$root timeout -k 10 200 python tools/perf/r4/uniform_ab.py 2>/dev/null | tail -1 | tee -a $out/uniform_ab.txt || exit 1
    done
done
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "uniform" > $out/uniform_tests.log 2>&1; echo "uniform tests rc=$?"; tail -3 $out/uniform_tests.log
