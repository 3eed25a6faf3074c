#!/bin/bash
# round 3, batch 18: the headline under rocprofv3 as the driver runs it (kernel timing on: both large-batch kernels appear, the
# chosen one with the timed launches) and with the one-tile kernel forced (counters)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3 gpurun_out/prof_r3_headline_default
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3_headline_default/trace -o trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-configs > gpurun_out/prof_r3_headline_default/bench.json 2> gpurun_out/prof_r3_headline_default/trace.err
cut -c1-200 gpurun_out/prof_r3_headline_default/trace/*kernel_stats.csv | head -6
grep '^{' gpurun_out/prof_r3_headline_default/bench.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['roofline']['kernel'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d['roofline']['large_batch_timing'])"
MEMB_HIP_PERSISTENT=0 timeout -k 10 500 bash tools/perf/prof.sh r3_headline_onetile "decode_trained<" > gpurun_out/r3/b18_prof_onetile.log 2>&1; grep -E "^(trace|bench|traffic_over|hbm_|lds_conflict|read_latency|write_latency)" gpurun_out/r3/b18_prof_onetile.log
