"""Helpers of bench.py that are not the benchmark: HIP-event timing, the PMC counter passes behind roofline.traffic,
the hash that ties profiles/hbm_traffic.json to a tree, writing the synthetic models. Nothing here imports oracle/."""
import hashlib
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BENCH = os.path.join(REPO, 'bench.py')


class Timer:
    """Per-launch device time from HIP events on torch's current stream (the stream the kernels are enqueued on:
    Reader.rows_embedding_device passes it into the C ABI)."""

    def __init__(self, torch):
        self.torch = torch

    def _run_in(self, call, run_in_ms):
        """~run_in_ms of the same launches without a gap: the part's power state needs that long to settle."""
        torch = self.torch
        call()
        torch.cuda.synchronize()
        begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        begin.record()
        call()
        end.record()
        torch.cuda.synchronize()
        one = max(begin.elapsed_time(end), 1e-3)
        return max(3, min(2000, int(run_in_ms / one) + 1))

    def launches(self, call, count, run_in_ms=20.0):
        """Sorted per-launch times of `count` launches, each between its own pair of events."""
        torch = self.torch
        events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(count)]
        for _ in range(self._run_in(call, run_in_ms)):
            call()
        for begin, end in events:
            begin.record()
            call()
            end.record()
        torch.cuda.synchronize()
        return sorted(begin.elapsed_time(end) for begin, end in events)

    def bursts(self, call, count, repeats=5, run_in_ms=20.0):
        """`repeats` bursts of `count` back-to-back launches, ONE event pair per burst: sorted averages per launch.
        (An event pair around every launch adds 4-5 us of its own: kernels below 0.2 ms are quoted this way.)"""
        torch = self.torch
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(repeats + 1)]
        for _ in range(self._run_in(call, run_in_ms)):
            call()
        marks[0].record()
        for k in range(repeats):
            for _ in range(count):
                call()
            marks[k + 1].record()
        torch.cuda.synchronize()
        return sorted(marks[k].elapsed_time(marks[k + 1]) / count for k in range(repeats))

    def median_ms(self, call, launches):
        """(median ms, how) of one launch: per-launch events, or bursts for kernels below 0.2 ms."""
        ms = self.launches(call, launches)
        if ms[len(ms) // 2] >= 0.2:
            return ms[len(ms) // 2], 'per-launch HIP events'
        averages = self.bursts(call, max(launches, 50))
        return averages[len(averages) // 2], 'bursts of back-to-back launches, one HIP event pair per burst'


def sources_sha16():
    """Hash of the kernel sources this tree builds from (memb_amd/csrc/*, include/memb_hip.h): profiles/hbm_traffic.json
    carries the hash of the tree its counters were taken on; counters of another tree are reported as stale, not as traffic."""
    digest = hashlib.sha256()
    csrc = os.path.join(REPO, 'memb_amd', 'csrc')
    for path in sorted(os.path.join(csrc, name) for name in os.listdir(csrc)) + [os.path.join(REPO, 'include', 'memb_hip.h')]:
        if os.path.isfile(path):
            digest.update(os.path.basename(path).encode())
            with open(path, 'rb') as f:
                digest.update(f.read())
    return digest.hexdigest()[:16]


def recorded_traffic():
    """profiles/hbm_traffic.json (rocprofv3 --pmc passes, tools/perf/profiles.sh) when it was taken on THIS tree's kernels."""
    path = os.path.join(REPO, 'profiles', 'hbm_traffic.json')
    if not os.path.exists(path):
        return {}, 'none recorded'
    with open(path) as f:
        recorded = json.load(f)
    if recorded.get('_sources_sha16') != sources_sha16():
        return {}, 'stale'
    return recorded, 'profiles/hbm_traffic.json'


def live_traffic(workload, kernel_name, cache_dir, timeout=90):
    """HBM bytes per launch of the timed kernel from the PMC counters, collected in THIS run: two child processes,
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, no trace domain: MI355X_MICROARCH.md), each
    over `python3 bench.py --workload <this one> --steps 3` with everything but the timed step off. FETCH_SIZE (KB)
    counts 64 B per 128-byte request of wide reads on gfx950 and is doubled, WRITE_SIZE (KB) is exact.
    Returns (bytes, source) or (None, why not)."""
    import csv
    import glob
    import shutil
    import tempfile
    profiler = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if profiler is None:
        return None, 'rocprofv3 not found'
    if any(name.startswith(('ROCPROF', 'ROCPROFILER', 'ROCP_')) for name in os.environ) or 'rocprofiler' in os.environ.get('LD_PRELOAD', ''):
        return None, 'this run is itself under a profiler'
    readings = {}
    scratch = tempfile.mkdtemp(prefix='memb_bench_pmc_', dir='/tmp')
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = os.path.join(scratch, counter)
            command = [profiler, '--pmc', counter, '--output-format', 'csv', '-d', out, '-o', 'pmc', '--',
                       sys.executable, BENCH, '--workload', workload, '--steps', '3', '--warmup', '1',
                       '--no-configs', '--no-cpu-baseline', '--no-live-traffic', '--cache-dir', cache_dir]
            try:
                done = subprocess.run(command, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp', MEMB_BENCH_PREBUILT='1'),
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
            except subprocess.TimeoutExpired:
                return None, 'rocprofv3 --pmc {} pass timed out'.format(counter)
            if done.returncode != 0:
                return None, 'rocprofv3 --pmc {} pass failed ({})'.format(counter, done.returncode)
            values = []
            for path in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        if row.get('Counter_Name') == counter and kernel_name in row.get('Kernel_Name', ''):
                            values.append(float(row['Counter_Value']))
            if not values:
                return None, 'no {} readings for {}'.format(counter, kernel_name)
            readings[counter] = sum(values) / len(values)
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    return int(round(2.0 * readings['FETCH_SIZE'] * 1024 + readings['WRITE_SIZE'] * 1024)), \
        'live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this run (2 x FETCH_SIZE KB + WRITE_SIZE KB)'


def prebuild_models(synthetic, models, workers=3):
    """Write the synthetic models this run needs and the box does not have yet, a few at a time."""
    from concurrent.futures import ThreadPoolExecutor
    start = time.time()
    missing = [m for m in models if not os.path.exists(synthetic.cached_model_path(*m))]
    if missing:
        with ThreadPoolExecutor(max_workers=min(workers, len(missing))) as pool:
            list(pool.map(lambda m: synthetic.cached_model(*m), missing))
    return time.time() - start if missing else 0.0


# --------------------------------------------------------------------------------------------
# launching the ranks
# --------------------------------------------------------------------------------------------

def kfd_gpu_count(topology='/sys/class/kfd/kfd/topology/nodes'):
    """GPUs of this node as the kernel driver lists them (topology nodes with SIMDs). sysfs only -- no HIP, no torch."""
    try:
        nodes = os.listdir(topology)
    except OSError:
        return None
    count = 0
    for node in nodes:
        try:
            with open(os.path.join(topology, node, 'properties')) as f:
                for line in f:
                    fields = line.split()
                    if len(fields) == 2 and fields[0] == 'simd_count' and int(fields[1]) > 0:
                        count += 1
        except (OSError, ValueError):
            continue
    return count


def hip_runtime_mapped():
    """Has this process mapped a HIP / HSA runtime library (the first step of touching the GPU)?"""
    with open('/proc/self/maps') as f:
        return any('libamdhip64' in line or 'libhsa-runtime64' in line for line in f)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of this process, which never
    touches the GPU (imports neither torch nor memb_amd, counts GPUs from sysfs, checks /proc/self/maps)."""
    import socket
    import build_native
    build_native.build_all()
    rehearsal = os.environ.get('MEMB_BENCH_REHEARSAL') in ('1', 'cpu')
    available = kfd_gpu_count()
    if not rehearsal and (available is None or 0 < available < args.gpus):
        raise SystemExit('--gpus {}: this node has {} GPU(s)'.format(args.gpus, available or 0))
    with socket.socket() as probe:
        probe.bind(('127.0.0.1', 0))
        port = probe.getsockname()[1]
    command = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), BENCH] + \
        [argument for argument in sys.argv[1:] if argument != '--dry-launch']
    launcher = {'started_by': 'bench.py (child processes)', 'parent_mapped_hip_runtime': hip_runtime_mapped(),
                'gpus_in_kfd_topology': available, 'parent_imported_torch': 'torch' in sys.modules}
    if launcher['parent_mapped_hip_runtime']:
        raise SystemExit('bench.py: the launching process has a HIP runtime mapped; refusing to start GPU ranks from it')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MEMB_BENCH_PREBUILT='1', MEMB_BENCH_LAUNCHER=json.dumps(launcher))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    if args.dry_launch:
        print(json.dumps({'launcher': launcher, 'command': command}))
        return 0
    return subprocess.run(command, env=env).returncode
