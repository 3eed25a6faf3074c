"""bench.py's output contract on small models: the single-GPU line with its per-configuration
array, and the N > 1 path started the way the driver may start it -- `python bench.py --gpus 2`
with no launcher -- as a rehearsal on one device (both ranks on cuda:0, gloo rendezvous)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

gpu = pytest.mark.gpu

CONTRACT_KEYS = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline')


def run_bench(arguments, tmp_path, **extra_env):
    env = dict(os.environ, MEMB_BENCH_CACHE=str(tmp_path / 'models'), MEMB_BENCH_DETAIL=str(tmp_path / 'bench_detail.json'), **extra_env)
    result = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py')] + arguments, env=env,
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert result.returncode == 0, result.stderr[-3000:]
    # what the driver keeps: the last 8 081 characters of stdout. The record must be whole in there, and alone on stdout.
    lines = result.stdout.splitlines()
    # (a gloo rehearsal's rendezvous chatter -- "[Gloo] Rank 3 is connected to ...", lines of several ranks interleaved --
    # comes first; RCCL runs have none)
    assert len([line for line in lines if line.startswith('{')]) == 1 and lines[-1].startswith('{'), result.stdout[-2000:]
    assert len(lines[-1]) < 8000, len(lines[-1])
    line = json.loads(result.stdout[-8081:].splitlines()[-1])
    assert line == json.loads(lines[-1])
    with open(tmp_path / 'bench_detail.json') as f:   # everything measured, verbose: the side file (also on stderr)
        detail = json.load(f)
    assert detail['metric'] == line['metric'] and 'detail: {' in result.stderr
    line['_detail'] = detail
    return line


def test_the_record_fits_the_drivers_window():
    """Round 5's line had grown to 20 KB and the driver, which keeps the last 8 081 characters of stdout, could not parse it.
    The record is built by bench.compact_line: worst-case field values (eight ranks, eight configurations, the longest kernel
    and workload names, a traffic source with a reason) must stay below 8 000 characters and keep `roofline` and
    `cpu_baseline` whole."""
    sys.path.insert(0, REPO)
    import bench
    long_kernel = 'decode_union_split<false, true, false, true> / decode_records_persistent<false, 2, false>'
    per_rank = [{'rank': r, 'device': r, 'batch': 2196017, 'kernel_avg_ms': 0.53123, 'reader_open_s': 12.345,
                 'device_bytes': 987654321012, 'word_index_bytes': 162345678} for r in range(8)]
    configs = [{'workload': 'glove840b-300d-2bit-fullvocab (BASELINE.json configs[3], one GPU: the whole dump)', 'kernel': long_kernel,
                'batch': 2196017, 'kernel_ms': 0.123456789, 'frac': 0.123456789, 'repeated_buffer_frac': 0.654321987,
                'algorithmic_bytes': 2940965007, 'timing': 'x' * 200, 'traffic': 2995617568, 'traffic_over_algorithmic': 1.0185832,
                'traffic_source': 'profiles/hbm_traffic.json', 'parity': 'bit-exact (20000 sampled rows)'} for _ in range(10)]
    strong = {'workload': 'glove840b-300d-2bit-fullvocab (BASELINE.json configs[3]): ONE dump split over the ranks, no collective',
              'scaling': 'strong', 'n_gpus': 8, 'ranks_seen': 8, 'steps': 20,
              'kernel_only': {'value': 3.3e10, 'unit': 'embeddings/s', 'ms_per_step': 0.0664321},
              'with_d2h': {'value': 3.3e8, 'unit': 'embeddings/s', 'ms_per_step': 6.64321},
              'host_gather': {'value': 3.3e8, 'unit': 'embeddings/s', 'ms_per_step': 6.64321, 'parity_rank0': 'bit-exact (5000 sampled rows)'},
              'per_rank': [{'rank': r, 'device': r, 'rows': [274503 * r, 274503 * (r + 1)], 'kernel_avg_ms': 0.06643, 'parity': 'bit-exact (5000 sampled rows)'}
                           for r in range(8)]}
    result = {
        'metric': 'embeddings/sec (and HBM GB/s vs roofline), 300-dim 4-bit batch lookup', 'value': 3.3123456789e10, 'unit': 'embeddings/s',
        'n_gpus': 8, 'steps': 20, 'warmup': 5, 'ms_per_step': 0.53123456789, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'u32', 'data': 'synthetic',
        'config': {'workload': 'glove840b-300d-4bit-fullvocab', 'vocabulary': 2196017, 'dim': 300, 'storage': 'trained', 'bits_per_weight': 4,
                   'batch_per_gpu': 2196017, 'batch': 'y' * 220, 'vectors': 'N(0, 0.4^2) seed 1234, written by memb_amd.Builder',
                   'parallelism': 'batch shards, model replicated per GPU, no collective'},
        'roofline': {'bound': 'hbm', 'achieved': 5573.852655525784, 'peak': 8000.0, 'unit': 'GB/s', 'frac': 0.696731581940723,
                     'traffic': 2995617568, 'traffic_source': 'z' * 200, 'traffic_over_algorithmic': 1.018583206828343, 'kernel': long_kernel,
                     'kernel_avg_ms': 0.5276359438896179, 'kernel_min_ms': 0.5117239952087402, 'kernel_median_ms': 0.5258039832115173,
                     'kernel_timing': 't' * 160, 'algorithmic_bytes_per_launch': 2940965007, 'algorithmic_bytes_per_word': 1339.2268853109972,
                     'kernel_ms_in_launch_order': [0.5179] * 200, 'box_fill': {'what': 'w' * 300, 'ms': 0.455, 'GBps': 5790.1}},
        'cpu_baseline': {'value': 23123456.789, 'unit': 'embeddings/s', 'cores': 256, 'kind': 'reference', 'sample': 's' * 200},
        'parity_vs_cpu_checker': 'bit-exact', 'ranks_seen': 8, 'rehearsal': 'r' * 110,
        'launcher': {'started_by': 'bench.py (child processes)', 'parent_mapped_hip_runtime': False, 'gpus_in_kfd_topology': 8, 'parent_imported_torch': False},
        'per_rank': per_rank, 'strong_scaling': strong, 'configs': configs, 'kernel_embeddings_per_s': 4.127e9,
        'geometry': {'waves_per_block': 8, 'tiles_per_wavefront': 1, 'lanes_per_word': 8, 'lds_bytes_per_block': 31232, 'row_bytes': 160},
        'sources_sha16': '0123456789abcdef', 'model_build_s': 123.45, 'reader_open_s': 12.345, 'extras': {'anything': 'e' * 20000},
    }
    # (configurations are measured at N = 1 only; the strong-scaling leg and eight per-rank summaries at N > 1)
    single = dict(result, n_gpus=1, ranks_seen=1, per_rank=per_rank[:1], strong_scaling=None)
    text = bench.compact_line(single)
    assert len(text) < 8000 and '\n' not in text, len(text)
    line = json.loads(text)
    assert len(line['configs']) == 10 and all(set(entry) <= set(bench.CONFIG_KEYS) for entry in line['configs'])
    text = bench.compact_line(dict(result, configs=None, cpu_baseline=None))
    assert len(text) < 8000 and '\n' not in text, len(text)
    line = json.loads(text)
    for key in CONTRACT_KEYS:
        assert key in line, key
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source', 'traffic_over_algorithmic', 'kernel', 'kernel_avg_ms',
                'algorithmic_bytes_per_launch'):
        assert key in line['roofline'], key
    assert json.loads(bench.compact_line(single))['cpu_baseline']['cores'] == 256
    assert 'extras' not in line and 'kernel_ms_in_launch_order' not in line['roofline']
    # a record that would not fit loses precision and optional keys, never the contract's: thirty ranks and forty configurations
    crowded = dict(result, per_rank=per_rank * 4, configs=configs * 4, strong_scaling=dict(strong, per_rank=strong['per_rank'] * 4))
    text = bench.compact_line(crowded)
    assert len(text) < 8000, len(text)
    for key in CONTRACT_KEYS:
        assert key in json.loads(text), key
    assert len(line['per_rank']) == 8 and len(line['strong_scaling']['per_rank']) == 8 and line['ranks_seen'] == 8


def test_launcher_parent_never_touches_the_gpu(native, tmp_path):
    """`python bench.py --gpus 8` as the driver may start it: the parent that spawns the ranks must not have
    mapped a HIP runtime (it imports neither torch nor memb_amd; devices are counted from sysfs)."""
    result = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '8', '--small', '--dry-launch'],
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600,
                            env=dict(os.environ, MEMB_BENCH_REHEARSAL='1'))
    assert result.returncode == 0, result.stderr[-2000:]
    plan = json.loads(result.stdout.strip().splitlines()[-1])
    assert plan['launcher']['parent_mapped_hip_runtime'] is False
    assert plan['launcher']['parent_imported_torch'] is False
    command = plan['command']
    assert command[command.index('--nproc-per-node') + 1] == '8' and '--dry-launch' not in command
    assert command[command.index('--master-addr') + 1] == '127.0.0.1'


def test_eight_rank_plumbing_on_the_cpu(native, tmp_path):
    """`python bench.py --gpus 8` the way the driver's scaling run starts it, minus the GPUs: eight ranks started by
    the script itself, gloo rendezvous, host stand-ins for everything that would touch a device
    (MEMB_BENCH_REHEARSAL=cpu: bench.install_host_stand_ins). What it pins is the plumbing -- every rank reports,
    the strong-scaling split covers the vocabulary once, the line keeps its contract -- not a single number."""
    line = run_bench(['--gpus', '8', '--small', '--steps', '2', '--warmup', '1'], tmp_path,
                     MEMB_BENCH_REHEARSAL='cpu', OMP_NUM_THREADS='1')
    for key in CONTRACT_KEYS:
        assert key in line, key
    assert line['rehearsal'].startswith('cpu')
    assert line['n_gpus'] == 8 and line['ranks_seen'] == 8 and line['scaling'] == 'weak'
    assert [entry['rank'] for entry in line['per_rank']] == list(range(8))
    assert all(entry['batch'] == 50000 for entry in line['per_rank'])          # weak scaling: a full batch per rank
    assert line['launcher']['parent_mapped_hip_runtime'] is False and line['launcher']['parent_imported_torch'] is False
    assert line['parity_vs_cpu_checker'].startswith('bit-exact')
    strong = line['strong_scaling']
    assert strong['scaling'] == 'strong' and strong['ranks_seen'] == 8 and strong['n_gpus'] == 8
    spans = [entry['rows'] for entry in strong['per_rank']]
    assert spans[0][0] == 0 and spans[-1][1] == 50000
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))                  # disjoint and gap-free: [0, n) once
    assert all(stop - start == 6250 for start, stop in spans)                   # ceil(50000 / 8), src/reader.cpp:65
    assert all(entry['parity'].startswith('bit-exact') for entry in strong['per_rank'])
    assert strong['host_gather']['parity_rank0'].startswith('bit-exact')

    strong_main = run_bench(['--gpus', '8', '--small', '--steps', '2', '--warmup', '1', '--scaling', 'strong'],
                            tmp_path, MEMB_BENCH_REHEARSAL='cpu', OMP_NUM_THREADS='1')
    assert strong_main['scaling'] == 'strong' and strong_main['ranks_seen'] == 8
    assert sum(entry['batch'] for entry in strong_main['per_rank']) == 50000


def test_the_process_group_never_carries_data(native):
    """north_star: "no RCCL collective needed, only a host-side gather". Every call bench.py makes on
    torch.distributed: the rendezvous, barriers, ONE kind of reduction (MAX, of an elapsed time) and the gather of the
    per-rank summary dictionaries. No all_gather / all_to_all / broadcast / send of tensors -- also not in the package."""
    import re
    allowed = {'init_process_group', 'destroy_process_group', 'barrier', 'all_reduce', 'all_gather_object', 'ReduceOp'}
    source = open(os.path.join(REPO, 'bench.py')).read()
    calls = set(re.findall(r'\bdist\.([A-Za-z_]+)', source))
    assert calls <= allowed, calls - allowed
    assert calls >= {'init_process_group', 'barrier', 'all_reduce', 'all_gather_object'}
    reductions = re.findall(r'dist\.all_reduce\(([^\n]*)\)', source)
    assert reductions and all('ReduceOp.MAX' in arguments for arguments in reductions), reductions
    # the package: sharding.gather_rows is the one place with a collective, and it runs on a gloo (host) group
    for name in os.listdir(os.path.join(REPO, 'memb_amd')):
        if name.endswith('.py'):
            text = open(os.path.join(REPO, 'memb_amd', name)).read()
            used = set(re.findall(r'\bdist\.([a-z_]+)\(', text))
            if name == 'sharding.py':
                assert used <= {'get_backend', 'get_process_group_ranks', 'get_world_size', 'new_group', 'destroy_process_group', 'get_rank', 'get_global_rank', 'gather'}, used
            else:
                assert not used, (name, used)


def test_recorded_counters_are_reported_only_for_the_tree_they_were_taken_on(monkeypatch):
    """profiles/hbm_traffic.json carries the hash of the kernel sources its rocprofv3 passes ran on (collect_profiles.py);
    bench.py quotes it for `configs[*].traffic_over_algorithmic` only from a tree with the same hash -- `stale` otherwise."""
    sys.path.insert(0, REPO)
    import bench
    import bench_support
    with open(os.path.join(REPO, 'profiles', 'hbm_traffic.json')) as f:
        recorded = json.load(f)
    assert len(recorded['_sources_sha16']) == 16 and recorded['_commit']
    for workload in ('glove840b-300d-4bit-fullvocab', 'glove840b-300d-4bit-100k', 'fasttext2m-300d-6bit-fullvocab',
                     'glove840b-300d-2bit-fullvocab', 'union-concat-500k', 'uniform-8bit-500k'):
        assert recorded[workload] > 0, workload
    monkeypatch.setattr(bench_support, 'sources_sha16', lambda: recorded['_sources_sha16'])
    values, source = bench_support.recorded_traffic()
    assert source == 'profiles/hbm_traffic.json' and values['glove840b-300d-4bit-100k'] == recorded['glove840b-300d-4bit-100k']
    monkeypatch.setattr(bench_support, 'sources_sha16', lambda: '0' * 16)
    assert bench_support.recorded_traffic() == ({}, 'stale')
    # the hash covers the kernel sources and the header, nothing else
    monkeypatch.undo()
    first = bench_support.sources_sha16()
    assert first == bench.sources_sha16() and len(first) == 16


def test_gpus_are_counted_from_the_kfd_topology(tmp_path):
    sys.path.insert(0, REPO)
    import bench
    for node, simds in enumerate((0, 0, 1024, 1024, 1024)):
        directory = tmp_path / 'nodes' / str(node)
        directory.mkdir(parents=True)
        (directory / 'properties').write_text('cpu_cores_count {}\nsimd_count {}\nmem_banks_count 1\n'.format(64 if not simds else 0, simds))
    assert bench.kfd_gpu_count(str(tmp_path / 'nodes')) == 3
    assert bench.kfd_gpu_count(str(tmp_path / 'absent')) is None
    assert bench.hip_runtime_mapped() in (False, True)


@gpu
def test_single_gpu_line_and_configuration_array(native, tmp_path):
    line = run_bench(['--small', '--steps', '3', '--warmup', '1', '--extras'], tmp_path)
    for key in CONTRACT_KEYS:
        assert key in line, key
    assert line['n_gpus'] == 1 and line['steps'] == 3 and line['scaling'] == 'weak' and line['vs_baseline'] is None
    assert line['parity_vs_cpu_checker'] == 'bit-exact'
    roofline = line['roofline']
    assert roofline['bound'] == 'hbm' and roofline['peak'] == 8000.0 and roofline['unit'] == 'GB/s'
    assert abs(roofline['frac'] - roofline['achieved'] / roofline['peak']) < 1e-5
    assert roofline['kernel'].startswith('decode_')   # (which kernel depends on the batch size: 50 000 words here)
    for key in ('traffic', 'traffic_source', 'traffic_over_algorithmic', 'kernel_avg_ms', 'algorithmic_bytes_per_launch'):
        assert key in roofline, key
    assert line['cpu_baseline']['kind'] in ('reference', 'port') and line['cpu_baseline']['cores'] >= 1
    workloads = [entry['workload'] for entry in line['configs']]
    for index in range(5):
        assert any('configs[{}]'.format(index) in name for name in workloads), (index, workloads)
    for entry in line['configs']:
        assert entry['parity'].startswith('bit-exact'), entry
        assert entry['kernel_ms'] > 0 and entry['frac'] > 0 and set(entry) <= set(('workload', 'kernel', 'batch', 'kernel_ms', 'frac',
                                                                                  'repeated_buffer_frac', 'traffic_over_algorithmic', 'traffic_source', 'parity'))
    # full-size batches in random order beside the dumps; configs[1]'s own figure is the HBM-regime one, the cache-assisted one a sub-field
    assert sum('shuffled' in name for name in workloads) == 2, workloads
    config1 = next(entry for entry in line['configs'] if 'configs[1]' in entry['workload'])
    assert config1['repeated_buffer_frac'] > 0
    rank0 = line['per_rank'][0]
    assert rank0['reader_open_s'] > 0 and rank0['device_bytes'] > rank0['word_index_bytes'] > 0   # the index is staged with the reader
    # the verbose side: everything the record leaves out, and the --extras legs
    detail = line['_detail']
    assert len(detail['roofline']['kernel_ms_in_launch_order']) == 3 and detail['roofline']['box_fill']['ms'] > 0
    assert all(entry['algorithmic_bytes'] > 0 for entry in detail['configs'])
    extras = detail['extras']
    search = extras['word_search']
    assert len(search['batches']) == 3 and search['index']['word_index_keys'] == 50000
    for entry in search['batches']:
        assert entry['parity'].startswith('device == host search'), entry
        assert entry['host_ms'] > 0 and entry['device_ms'] > 0
    assert extras['host_api']['batch_seconds'] > 0 and len(extras['small_batches']) == 4
    assert extras['four_batches_of_100k_in_one_launch']['frac'] > 0


@gpu
def test_two_ranks_started_without_a_launcher(native, tmp_path):
    line = run_bench(['--gpus', '2', '--small', '--steps', '2', '--warmup', '1'], tmp_path, MEMB_BENCH_REHEARSAL='1')
    assert line['n_gpus'] == 2 and line['ranks_seen'] == 2 and len(line['per_rank']) == 2
    assert line['launcher']['parent_mapped_hip_runtime'] is False and line['launcher']['parent_imported_torch'] is False
    assert line['parity_vs_cpu_checker'].startswith('bit-exact')
    strong = line['strong_scaling']
    assert strong['scaling'] == 'strong' and strong['ranks_seen'] == 2
    first, second = strong['per_rank']
    assert first['rows'][0] == 0 and first['rows'][1] == second['rows'][0] and second['rows'][1] == 50000
    assert all(entry['parity'].startswith('bit-exact') for entry in strong['per_rank'])
    # (no ordering between the two on a rehearsal box: both ranks share one GPU and 50 000-word batches are all launch latency)
    assert strong['kernel_only']['value'] > 0 and strong['with_d2h']['value'] > 0
    assert strong['host_gather']['value'] > 0 and strong['host_gather']['parity_rank0'].startswith('bit-exact')

    strong_main = run_bench(['--gpus', '2', '--small', '--steps', '2', '--warmup', '1', '--scaling', 'strong'],
                            tmp_path, MEMB_BENCH_REHEARSAL='1')
    assert strong_main['scaling'] == 'strong' and strong_main['config']['workload'] == 'glove840b-300d-2bit-fullvocab'
    assert sum(entry['batch'] for entry in strong_main['per_rank']) == 50000


@gpu
def test_four_rank_rehearsal(native, tmp_path):
    # as many ranks as a one-GPU box allows next to the test process (at most 6 processes may use the card);
    # the 8-way split itself is covered on the CPU: tests/test_sharding_gloo.py, tests/test_gpu_full_size.py
    line = run_bench(['--gpus', '4', '--small', '--steps', '2', '--warmup', '1', '--no-cpu-baseline'], tmp_path,
                     MEMB_BENCH_REHEARSAL='1')
    assert line['n_gpus'] == 4 and line['ranks_seen'] == 4
    strong = line['strong_scaling']
    assert [entry['rows'] for entry in strong['per_rank']] == [[0, 12500], [12500, 25000], [25000, 37500], [37500, 50000]]
    assert all(entry['parity'].startswith('bit-exact') for entry in strong['per_rank'])
