#!/bin/bash
# round 3, batch 6: automatic kernel choice by batch size against the forced kernels; full bench line; builder phases
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export AB3_ROUNDS=4 AB3_REPS=20
V='general:persistent=2;pipeline=0,records:persistent=2;pipeline=1,onetile:persistent=0'
for bits in 4 2; do
AB3_BITS=$bits AB3=$V AB3_CASES=sorted,random,500k,250k,100k,50k,10k,1k,union timeout -k 10 500 python3 tools/perf/ab3.py > gpurun_out/r3/b6_auto_bits$bits.log 2>&1 || { tail -30 gpurun_out/r3/b6_auto_bits$bits.log; exit 1; }
echo "bits $bits"; sed -n '/^---/,$p' gpurun_out/r3/b6_auto_bits$bits.log | grep -v "A/A"
done
BT_DEVICE=0 MEMB_BUILDER_VERBOSE=1 timeout -k 10 300 python3 tools/perf/buildtime.py > gpurun_out/r3/b6_buildtime_device.log 2>&1; grep -v "^memb_hip" gpurun_out/r3/b6_buildtime_device.log | tail -30
timeout -k 10 900 python3 bench.py > gpurun_out/r3/b6_bench.json 2> gpurun_out/r3/b6_bench.err; tail -c 600 gpurun_out/r3/b6_bench.err
python3 - <<'PY'
import json
line=[l for l in open('gpurun_out/r3/b6_bench.json') if l.startswith('{')][-1]
d=json.loads(line)
print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'], d['model_build_s'], d['parity_vs_cpu_checker'])
for c in d['configs']: print('%-70s %-45s %.4f ms frac %.3f %s' % (c['workload'][:70], c['kernel'][:45], c['kernel_ms'], c['frac'], c['parity']))
print(d['cpu_baseline']); print(d['host_api']['batch_seconds'], d['host_api']['sample_seconds'])
PY
timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/r3/b6_pytest.log 2>&1; tail -5 gpurun_out/r3/b6_pytest.log
