#!/bin/bash
# Round 5: the rocprofv3 passes of BASELINE configs[1] alone (bench.py rotates four batches into four buffers for it now).
set -o pipefail
mkdir -p gpurun_out/r5_profiles
export MEMB_SYNTH_DEVICE=0
timeout -k 10 700 bash tools/perf/prof.sh r05_100k "decode_records_persistent" --workload glove840b-300d-4bit-100k > gpurun_out/r5_profiles/prof_r05_100k.txt 2>&1 || { tail -20 gpurun_out/r5_profiles/prof_r05_100k.txt; exit 1; }
grep "AverageNs\|traffic_over_algorithmic\|'frac'" gpurun_out/r5_profiles/prof_r05_100k.txt | head -5
