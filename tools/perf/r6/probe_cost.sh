#!/bin/bash
# Round 6: does looking at the batch's order cost the batch anything? Two shipped builds alternating as processes: the tree, and the tree
# compiled with -DMEMB_HIP_NO_ORDER_PROBE (build/noprobe: no noteBatchOrder call in decode_trained; the host then never sees an order).
set -o pipefail
out=gpurun_out/r6_probe_cost
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
for round in 1 2 3; do
    for root in "" build/noprobe; do
        MEMB_PACKAGE_ROOT=$root timeout -k 10 200 python tools/perf/r6/dumps.py 2>&1 | grep -v amdgpu.ids | tee -a $out/dumps.txt || exit 1
    done
done
