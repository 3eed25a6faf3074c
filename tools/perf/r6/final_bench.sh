#!/bin/bash
# The default bench.py as the driver runs it (the record on stdout), then once more with --extras (ceilings, word search, host API into the detail).
set -o pipefail
out=gpurun_out/r6_final_${1:-a}
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
start=$(date +%s)
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }
echo "default run: $(( $(date +%s) - start )) s, record $(wc -c < $out/bench.json) bytes"
cp gpurun_out/bench_detail.json $out/bench_detail.json
timeout -k 10 500 python bench.py --extras --no-configs > $out/extras.json 2> $out/extras.err || { tail -20 $out/extras.err; exit 1; }
cp gpurun_out/bench_detail.json $out/extras_detail.json
python - $out <<'P'
import json, sys
out = sys.argv[1]
line = json.load(open(out + '/bench.json'))
r = line['roofline']
print('headline value %.4g  frac %.4f  kernel_avg_ms %.4f  traffic x%s (%s)  cpu %.3g/s on %d cores  parity %s' % (
    line['value'], r['frac'], r['kernel_avg_ms'], r['traffic_over_algorithmic'], r['traffic_source'][:30], line['cpu_baseline']['value'], line['cpu_baseline']['cores'], line['parity_vs_cpu_checker']))
for c in line['configs']:
    print('%-72s %-45s %.4f ms %.3f %s x%s %s' % (c['workload'][:72], c['kernel'], c['kernel_ms'], c['frac'], c.get('repeated_buffer_frac', ''), c.get('traffic_over_algorithmic'), c['parity'][:9]))
extras = json.load(open(out + '/extras_detail.json'))['extras']
for key, value in extras['box_ceilings'].items():
    if isinstance(value, dict) and 'ms' in value:
        print('  ceiling %-42s %.4f ms' % (key, value['ms']))
print('  kernel over fastest pattern', extras['box_ceilings']['kernel_against_the_fastest_pattern'])
for b in extras['word_search']['batches']:
    print('  words %-34s host %.3f ms  device %.3f ms  packed %.3f ms  %s / %s' % (b['batch'], b['host_ms'], b['device_ms'], b['device_ms_from_packed_words'], b['parity'][:13], b['packed_parity']))
print('  small', [(s['batch'], round(s['us_per_launch'], 2), round(s['frac'], 3)) for s in extras['small_batches']], extras['four_batches_of_100k_in_one_launch'])
print('  host api', {k: v for k, v in extras['host_api'].items() if k in ('batch_seconds', 'sample_seconds', 'cpu_port_sample_seconds')})
P
