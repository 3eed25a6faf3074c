#!/bin/bash
# round 3, batch 17: traffic of the 2-bit dump (both kernels), final GPU suite and bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
timeout -k 10 500 bash tools/perf/prof.sh r3_2bit_persistent decode_trained_persistent --workload glove840b-300d-2bit-fullvocab > gpurun_out/r3/b17_prof_2bit_p.log 2>&1; grep -E "^(trace|bench|traffic_over|hbm_)" gpurun_out/r3/b17_prof_2bit_p.log
MEMB_HIP_PERSISTENT=0 timeout -k 10 500 bash tools/perf/prof.sh r3_2bit_onetile "decode_trained<" --workload glove840b-300d-2bit-fullvocab > gpurun_out/r3/b17_prof_2bit_o.log 2>&1; grep -E "^(trace|bench|traffic_over|hbm_)" gpurun_out/r3/b17_prof_2bit_o.log
timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/r3/b17_pytest.log 2>&1; tail -4 gpurun_out/r3/b17_pytest.log
timeout -k 10 900 python3 bench.py --cache-dir /tmp/memb_fresh_cache3 > gpurun_out/r3/b17_bench.json 2> gpurun_out/r3/b17_bench.err; tail -c 300 gpurun_out/r3/b17_bench.err
python3 - <<'PY'
import json
line=[l for l in open('gpurun_out/r3/b17_bench.json') if l.startswith('{')][-1]
d=json.loads(line)
print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'], 'model_build_s', d['model_build_s'], d['parity_vs_cpu_checker'])
for c in d['configs']: print('%-55s %-45s %.4f ms frac %.3f %s' % (c['workload'][:55], c['kernel'][:45], c['kernel_ms'], c['frac'], c['parity']))
for s in d['configs'][1]['small_batches_of_the_same_model']: print(s)
PY
