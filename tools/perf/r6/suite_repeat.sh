#!/bin/bash
# The -m gpu suite several times in a row (one process each), C-level stderr NOT captured by pytest (--capture=sys): an intermittent host-side abort
# hides its reason (glibc / libstdc++ write it to fd 2) inside pytest's fd capture otherwise. Stops at the first failure.
set -o pipefail
out=gpurun_out/r6_suite_repeat
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
for run in 1; do
    
    timeout -k 10 600 python -m pytest tests -x -q -m gpu --capture=sys > $out/run$run.txt 2>&1 || { echo "run $run FAILED"; grep -v "^  File\|^$" $out/run$run.txt | tail -30 | cut -c1-300; exit 1; }
    echo "run $run: $(tail -1 $out/run$run.txt)"
done
