#!/bin/bash
# Round 5, batch 3: the finer segment index of small batches (about sixteen lanes per word on row records) against eight
# lanes, by batch size, on the 4-, 6- and 2-bit models; its parity tests; then the full bench line.
set -o pipefail
out=gpurun_out/r5_batch3
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "finer_index or ragged or small_batches or every_kernel" > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -3 $out/tests.log
for model in "4 2196017" "6 1999995" "2 2196017"; do
    set -- $model
    # base = the rule (fine up to 16 k words); off = never; on = always
    AB3='off:fine_lanes=1,on:fine_lanes=2' AB3_BITS=$1 AB3_WORDS=$2 AB3_CASES=1k,5k,10k,16k,20k,30k,40k,50k,100k,rot100k AB3_ROUNDS=4 \
        timeout -k 10 400 python tools/perf/ab3.py > $out/fine_$1bit.txt 2>&1 || { tail -30 $out/fine_$1bit.txt; exit 1; }
    echo "$1-bit"; sed -n '/--- median/,$p' $out/fine_$1bit.txt | grep -v "^---\|A/A\|base2"
done
timeout -k 10 900 python bench.py > $out/bench.json 2> $out/bench.err || { tail -30 $out/bench.err; exit 1; }
python - <<'PY'
import json
line = json.loads(open('gpurun_out/r5_batch3/bench.json').read().strip().splitlines()[-1])
print('value', line['value'], 'frac', line['roofline']['frac'], 'kernel_avg_ms', line['roofline']['kernel_avg_ms'], 'open_s', line['reader_open_s'], 'device_bytes', line['geometry']['device_bytes'])
for entry in line['configs']:
    print('%-90s %-45s ms %.4f frac %.3f %s' % (entry['workload'][:90], entry['kernel'][:45], entry['kernel_ms'], entry['frac'], entry['parity'][:12]))
c1 = [e for e in line['configs'] if 'configs[1]' in e['workload']][0]
print('configs[1] repeated', c1['repeated_buffer']['frac'], 'many', json.dumps(c1['batches_in_one_launch']))
print(json.dumps(c1['small_batches_of_the_same_model']))
print(json.dumps(line['word_search'], indent=1))
PY
