"""gpurun_out/prof_<round>_<tag>/ (tools/perf/prof.sh) -> profiles/<round>_<tag>_{summary.json, kernel_stats.csv,
bench_under_rocprof.json} and profiles/hbm_traffic.json, stamped with the commit the passes ran on.

    python tools/perf/collect_profiles.py r06 <commit>

`_sources_sha16` = the hash of memb_amd/csrc/* + include/memb_hip.h at collection time (bench_support.sources_sha16):
bench.py reports these counters only from a tree with the same hash, `traffic_source: stale` otherwise.
"""
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_support import sources_sha16   # noqa: E402
WORKLOADS = {   # tag -> bench.py workload
    'headline': 'glove840b-300d-4bit-fullvocab', '100k': 'glove840b-300d-4bit-100k', 'union': 'union-concat-500k',
    '6bit': 'fasttext2m-300d-6bit-fullvocab', '2bit': 'glove840b-300d-2bit-fullvocab', 'uniform': 'uniform-8bit-500k'}


def main(round_tag, commit):
    traffic = {
        '_how': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --workload <name> '
                '--steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-live-traffic` (tools/perf/prof.sh '
                '{0}_<tag> <kernel>; tools/perf/profiles.sh), mean over the launches of the pass; profiles/{0}_*_summary.json. '
                'FETCH_SIZE (KB) = TCC_EA0_RDREQ x 64 B; on gfx950 it reports half of wide reads (MI355X_MICROARCH.md, HBM '
                'section), so read bytes = 2 x FETCH_SIZE x 1024. WRITE_SIZE x 1024 = the fp32 output exactly.'.format(round_tag),
        '_commit': commit, '_sources_sha16': sources_sha16(), '_round': int(round_tag.lstrip('r')), '_detail': {}}
    for tag, workload in WORKLOADS.items():
        source = os.path.join(REPO, 'gpurun_out', 'prof_{}_{}'.format(round_tag, tag))
        with open(os.path.join(source, 'summary.json')) as f:
            summary = json.load(f)
        if summary.get('hbm_traffic_bytes') is None:
            raise SystemExit('{}: no traffic in summary.json (a counter pass failed?)'.format(source))
        target = os.path.join(REPO, 'profiles', '{}_{}'.format(round_tag, tag))
        shutil.copy(os.path.join(source, 'summary.json'), target + '_summary.json')
        shutil.copy(os.path.join(source, 'kernel_stats.csv'), target + '_kernel_stats.csv')
        shutil.copy(os.path.join(source, 'bench.json'), target + '_bench_under_rocprof.json')
        traffic[workload] = int(round(summary['hbm_traffic_bytes']))
        traffic['_detail'][tag] = {
            'kernel': summary['kernel'],
            'read_bytes_corrected': summary['hbm_read_bytes_corrected'], 'write_bytes': summary['hbm_write_bytes'],
            'FETCH_SIZE_KB': summary['FETCH_SIZE'], 'WRITE_SIZE_KB': summary['WRITE_SIZE'],
            'traffic_over_algorithmic': summary.get('traffic_over_algorithmic'),
            'rocprof_kernel_average_ns': float(summary['trace']['AverageNs']), 'rocprof_calls': int(summary['trace']['Calls']),
            'lds_conflict_share': summary.get('lds_conflict_share')}
        print('{:9s} {:>10.1f} ns x {:4d}  traffic x{:.4f}'.format(
            tag, float(summary['trace']['AverageNs']), int(summary['trace']['Calls']), summary.get('traffic_over_algorithmic') or 0))
    with open(os.path.join(REPO, 'profiles', 'hbm_traffic.json'), 'w') as f:
        json.dump(traffic, f, indent=1)
        f.write('\n')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
