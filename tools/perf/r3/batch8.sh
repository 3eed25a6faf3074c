#!/bin/bash
# round 3, batch 8: the large-batch kernel settled by timing on first use (base) against both forced kernels;
# a full bench line on a fresh model cache (model_build_s); GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export AB3_ROUNDS=4 AB3_REPS=20
V='general:persistent=2;pipeline=0,onetile:persistent=0'
for bits in 4 2 6; do
MEMB_HIP_VERBOSE=1 AB3_BITS=$bits AB3=$V AB3_CASES=sorted,random,250k,100k,10k timeout -k 10 400 python3 tools/perf/ab3.py > gpurun_out/r3/b8_autotune_bits$bits.log 2>&1 || { tail -30 gpurun_out/r3/b8_autotune_bits$bits.log; exit 1; }
echo "bits $bits"; grep "large batches" gpurun_out/r3/b8_autotune_bits$bits.log | head -3; sed -n '/^---/,$p' gpurun_out/r3/b8_autotune_bits$bits.log | grep -v "A/A"
done
timeout -k 10 900 python3 bench.py --cache-dir /tmp/memb_fresh_cache > gpurun_out/r3/b8_bench.json 2> gpurun_out/r3/b8_bench.err; tail -c 400 gpurun_out/r3/b8_bench.err
python3 - <<'PY'
import json
line=[l for l in open('gpurun_out/r3/b8_bench.json') if l.startswith('{')][-1]
d=json.loads(line)
print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'], 'model_build_s', d['model_build_s'], d['model_writer'], d['parity_vs_cpu_checker'])
for c in d['configs']: print('%-70s %-45s %.4f ms frac %.3f %s' % (c['workload'][:70], c['kernel'][:45], c['kernel_ms'], c['frac'], c['parity']))
print(d['cpu_baseline']['value'], d['host_api']['batch_seconds'], d['host_api']['sample_seconds'])
PY
timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/r3/b8_pytest.log 2>&1; tail -5 gpurun_out/r3/b8_pytest.log
