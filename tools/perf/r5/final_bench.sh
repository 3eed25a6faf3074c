#!/bin/bash
# the default bench line as the driver runs it (+ the smoke entry point), kept under gpurun_out/
set -o pipefail
out=gpurun_out/r5_final
mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1 || { tail -20 $out/smoke.txt; exit 1; }
tail -1 $out/smoke.txt
start=$(date +%s)
timeout -k 10 900 python bench.py > $out/bench.json 2> $out/bench.err || { tail -30 $out/bench.err; exit 1; }
echo "bench.py wall: $(( $(date +%s) - start )) s"
python - <<'PY'
import json
line = json.loads(open('gpurun_out/r5_final/bench.json').read().strip().splitlines()[-1])
r = line['roofline']
print('value %.4g frac %.4f kernel_avg_ms %.4f traffic x%.4f open_s %.3f' % (line['value'], r['frac'], r['kernel_avg_ms'], r['traffic_over_algorithmic'] or 0, line['reader_open_s']))
print(r['traffic_source'][:120])
for entry in line['configs']:
    extra = ''
    if 'with_random_order_hint' in entry:
        extra = ' | hinted %.4f ms frac %.3f' % (entry['with_random_order_hint']['kernel_ms'], entry['with_random_order_hint']['frac'])
    print('%-88s ms %.4f frac %.3f%s' % (entry['workload'][:88], entry['kernel_ms'], entry['frac'], extra))
c1 = [e for e in line['configs'] if 'configs[1]' in e['workload']][0]
print('configs[1] repeated %.3f | 4 in one launch frac %.3f (x%.2f per batch)' % (c1['repeated_buffer']['frac'], c1['batches_in_one_launch']['frac'], c1['batches_in_one_launch']['against_one_batch_per_launch']))
print([(s['batch'], round(s['us_per_launch'], 2), s['lanes_per_word']) for s in c1['small_batches_of_the_same_model']])
for b in line['word_search']['batches']:
    print('%-36s host %.3f ms device %.3f ms x%.1f' % (b['batch'], b['host_ms'], b['device_ms'], b['speedup']))
print('host_api batch %.1f ms, 100k %.2f ms; cpu_baseline %.3g emb/s on %d cores' % (line['host_api']['batch_seconds'] * 1e3, line['host_api']['sample_seconds'] * 1e3, line['cpu_baseline']['value'], line['cpu_baseline']['cores']))
c = r['box_ceilings']
print('ceilings: linear %.4f tile %.4f +seq %.4f +rand %.4f union %.4f | two tiles +seq %.4f +rand %.4f | kernel / fastest pattern %.3f (%s)' % (
    c['linear_fill']['ms'], c['tile_fill']['ms'], c['tile_fill_sequential_records']['ms'], c['tile_fill_random_records']['ms'], c['union_tile_fill_random_records']['ms'],
    c['two_tiles_sequential_records']['ms'], c['two_tiles_random_records']['ms'], c['kernel_against_the_fastest_pattern']['kernel_over_pattern'], c['kernel_against_the_fastest_pattern']['pattern']))
PY
