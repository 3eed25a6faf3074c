#!/bin/bash
# round 3, batch 42: the final tree's headline under rocprofv3 exactly as the driver would run it (default bench.py
# arguments apart from the CPU baseline and the configs), GPU suite first
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3 gpurun_out/prof_r3_headline_final
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r3/b42_pytest.log 2>&1 || { tail -30 gpurun_out/r3/b42_pytest.log; exit 1; }
tail -2 gpurun_out/r3/b42_pytest.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3_headline_final/trace -o trace -- python3 bench.py --no-cpu-baseline --no-configs > gpurun_out/prof_r3_headline_final/bench.json 2> gpurun_out/prof_r3_headline_final/trace.err
cut -c1-220 gpurun_out/prof_r3_headline_final/trace/*kernel_stats.csv | head -6
grep '^{' gpurun_out/prof_r3_headline_final/bench.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['roofline']['kernel'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d['roofline']['large_batch_timing'], d['steps'], d['warmup'])"
