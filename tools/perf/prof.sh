#!/bin/bash
# usage: tools/perf/prof.sh <tag> <kernel name substring> <bench.py arguments...>   (repo root, GPU box)
# One workload of bench.py under rocprofv3: kernel trace + stats, then the counter groups one pass each
# (never combined with a trace, FETCH_SIZE and WRITE_SIZE in passes of their own: MI355X_MICROARCH.md),
# summarised into gpurun_out/prof_<tag>/summary.json. Example:
#   tools/perf/prof.sh r04_union decode_union_split --workload union-concat-500k
# The trace pass also holds the box-ceiling kernels of tools/perf/ceilings.hip (bench.py runs them on the headline workload).
tag=$1; kernel=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag; mkdir -p $out
long="bench.py $* --no-configs --no-cpu-baseline --steps 10 --warmup 3"
short="bench.py $* --no-configs --no-cpu-baseline --no-ceilings --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- python3 $long > $out/bench.json 2> $out/trace.err || exit 1
pass() {   # name, counters...
    name=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o pmc -- python3 $short > /dev/null 2> $out/$name.err || echo "pass $name failed" >&2
}
pass pmc1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS
pass pmc2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR
pass pmc3 GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM
pass pmc4 TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pass tcc2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
python3 - "$out" "$kernel" <<'PY'
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
PY
cp $out/trace/*kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
