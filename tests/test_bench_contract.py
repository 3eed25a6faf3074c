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
    env = dict(os.environ, MEMB_BENCH_CACHE=str(tmp_path / 'models'), **extra_env)
    result = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py')] + arguments, env=env,
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert result.returncode == 0, result.stderr[-3000:]
    lines = [line for line in result.stdout.splitlines() if line.startswith('{')]
    assert len(lines) == 1, result.stdout[-2000:]
    return json.loads(lines[0])


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
    line = run_bench(['--small', '--steps', '3', '--warmup', '1'], tmp_path)
    for key in CONTRACT_KEYS:
        assert key in line, key
    assert line['n_gpus'] == 1 and line['steps'] == 3 and line['scaling'] == 'weak' and line['vs_baseline'] is None
    assert line['parity_vs_cpu_checker'] == 'bit-exact'
    roofline = line['roofline']
    assert roofline['bound'] == 'hbm' and roofline['peak'] == 8000.0 and roofline['unit'] == 'GB/s'
    assert abs(roofline['frac'] - roofline['achieved'] / roofline['peak']) < 1e-12
    assert roofline['kernel'].startswith('decode_')   # (which kernel depends on the batch size: 50 000 words here)
    assert line['cpu_baseline']['kind'] in ('reference', 'port') and line['cpu_baseline']['cores'] >= 1
    workloads = [entry['workload'] for entry in line['configs']]
    for index in range(5):
        assert any('configs[{}]'.format(index) in name for name in workloads), (index, workloads)
    for entry in line['configs']:
        assert entry['parity'].startswith('bit-exact'), entry
        assert entry['kernel_ms'] > 0 and entry['algorithmic_bytes'] > 0
    # round 5: full-size batches in random order beside the dumps; configs[1]'s own figure is the HBM-regime one, the
    # cache-assisted one a sub-field; several batches in one launch; the word search on the device
    assert sum('shuffled' in name for name in workloads) == 2, workloads
    for entry in line['configs']:
        if 'shuffled' in entry['workload']:
            assert entry['with_random_order_hint']['parity'].startswith('bit-exact') and entry['with_random_order_hint']['kernel_ms'] > 0
    config1 = next(entry for entry in line['configs'] if 'configs[1]' in entry['workload'])
    assert config1['frac_is'].startswith('HBM regime') and config1['repeated_buffer']['frac'] > 0
    assert config1['batches_in_one_launch']['parity'].startswith('bit-exact'), config1['batches_in_one_launch']
    assert config1['batches_in_one_launch']['batches'] == 4 and config1['batches_in_one_launch']['frac'] > 0
    union = next(entry for entry in line['configs'] if 'configs[4]' in entry['workload'])
    assert union['from_words']['parity'].startswith('bit-exact') and union['from_words']['ms'] > 0
    search = line['word_search']
    assert len(search['batches']) == 3 and search['index']['word_index_keys'] == 50000
    for entry in search['batches']:
        assert entry['parity'].startswith('device == host search'), entry
        assert entry['host_ms'] > 0 and entry['device_ms'] > 0
    rank0 = line['per_rank'][0]
    assert rank0['reader_open_s'] > 0 and rank0['device_bytes'] > rank0['word_index_bytes'] > 0   # the index is staged with the reader


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
