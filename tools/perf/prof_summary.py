"""Summary of one tools/perf/prof.sh run: counters of the kernel named by a substring, averaged over the launches of each
pass, the kernel trace line, the bench line; writes <out>/summary.json.   python3 tools/perf/prof_summary.py <out dir> <kernel substring>"""
import csv, collections, glob, json, sys
out, kernel = sys.argv[1], sys.argv[2]
summary = {'kernel_filter': kernel}
for path in sorted(glob.glob(out + '/*/*counter_collection.csv')):
    values = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if kernel in row['Kernel_Name'] and ', 3, ' not in row['Kernel_Name']:
            values[row['Counter_Name']].append(float(row['Counter_Value']))
            summary['kernel'] = row['Kernel_Name']
            for key in ('VGPR_Count', 'SGPR_Count', 'LDS_Block_Size', 'Scratch_Size', 'Grid_Size', 'Workgroup_Size'):
                if row.get(key):
                    summary[key] = row[key]
    for name, series in values.items():
        summary[name] = sum(series) / len(series)
        summary.setdefault('_launches', {})[name] = len(series)
for row in csv.DictReader(open(glob.glob(out + '/trace/*kernel_stats.csv')[0])):
    if kernel in row['Name']:
        summary['trace'] = {k: row[k] for k in ('Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'MinNs', 'MaxNs') if k in row}
        break
try:
    line = [l for l in open(out + '/bench.json') if l.startswith('{')][-1]
    bench = json.loads(line)
    summary['bench'] = {'workload': bench['config']['workload'], 'kernel_avg_ms_hip_events': bench['roofline']['kernel_avg_ms'],
                        'algorithmic_bytes': bench['roofline']['algorithmic_bytes_per_launch'], 'frac': bench['roofline']['frac'],
                        'parity': bench['parity_vs_cpu_checker']}
except Exception as error:
    summary['bench_error'] = str(error)
if summary.get('FETCH_SIZE') is not None and summary.get('WRITE_SIZE') is not None:
    # gfx950: FETCH_SIZE (KB) counts 64 B per 128-B request of wide reads -> doubled; WRITE_SIZE (KB) is exact
    summary['hbm_read_bytes_corrected'] = 2 * summary['FETCH_SIZE'] * 1024
    summary['hbm_write_bytes'] = summary['WRITE_SIZE'] * 1024
    summary['hbm_traffic_bytes'] = summary['hbm_read_bytes_corrected'] + summary['hbm_write_bytes']
    if 'bench' in summary:
        summary['traffic_over_algorithmic'] = summary['hbm_traffic_bytes'] / summary['bench']['algorithmic_bytes']
if summary.get('SQ_LDS_IDX_ACTIVE'):
    summary['lds_conflict_share'] = summary['SQ_LDS_BANK_CONFLICT'] / summary['SQ_LDS_IDX_ACTIVE']
if summary.get('TCP_TCC_READ_REQ_sum'):
    summary['read_latency_cycles'] = summary['TCP_TCC_READ_REQ_LATENCY_sum'] / summary['TCP_TCC_READ_REQ_sum']
    summary['write_latency_cycles'] = summary['TCP_TCC_WRITE_REQ_LATENCY_sum'] / max(summary['TCP_TCC_WRITE_REQ_sum'], 1)
json.dump(summary, open(out + '/summary.json', 'w'), indent=1, sort_keys=True)
for key in sorted(summary):
    print('%-32s %s' % (key, summary[key]))
