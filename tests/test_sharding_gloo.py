"""The N > 1 path: two processes (gloo), each takes its slice of the batch, slices
are gathered on the host and must equal the unsharded result. Every rank looks its
slice up with the product's own Reader and checks it against the CPU checker: in
the CPU suite a Reader on the host path (device='cpu'), in the GPU suite a Reader
on the device LOCAL_RANK picks (cuda:0 for both ranks on a one-GPU box)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import GOLDEN, REPO


def test_shard_range_covers_batch_like_reference_thread_split(native):
    from memb_amd.sharding import shard_range
    for count in (0, 1, 5, 1024, 1025, 2196017):
        for world in (1, 2, 3, 4, 8):
            ranges = [shard_range(count, r, world) for r in range(world)]
            job = (count + world - 1) // world  # reference src/reader.cpp:65
            assert ranges[0][0] == 0 and ranges[-1][1] == count
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            assert all(stop - start <= job for start, stop in ranges)
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, {repo!r})
    import oracle
    from memb_amd.sharding import shard_of, shard_range, gather_rows

    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    checker = oracle.OracleReader({model!r}, 1)
    import memb_amd
    if {use_hip!r}:
        devices = memb_amd.hip_device_count()
        assert devices >= 1
        reader = memb_amd.Reader({model!r}, device=int(os.environ['LOCAL_RANK']) % devices)
    else:
        reader = memb_amd.Reader({model!r}, device='cpu')
    keys = checker.keys()
    batch = [keys[(7 * i) % len(keys)] if i % 9 else 'missing-%d' % i for i in range({count})]
    mine = shard_of(batch, rank, world)
    local = reader.batch_embedding(mine)
    assert np.array_equal(local.view(np.uint32), checker.batch_embedding(mine).view(np.uint32))
    if {force_host_group!r}:
        # what gather_rows does under an `nccl` default group: a gloo group of its own for the host buffers
        import memb_amd.sharding as sharding
        sharding._host_groups[None] = sharding.host_group(None, force_new=True)
        assert dist.get_backend(sharding._host_groups[None]) == 'gloo'
        original = sharding.host_group
        sharding.host_group = lambda group=None, force_new=False: sharding._host_groups[None]
    full = gather_rows(local, len(batch), dst=0)
    if rank == 0:
        assert np.array_equal(full.view(np.uint32), checker.batch_embedding(batch).view(np.uint32))
        np.save({out!r}, full)
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()
''')


def run_two_ranks(tmp_path, use_hip, count, port, ranks=2, force_host_group=False):
    model = os.path.join(GOLDEN, 'synthetic_4bit.bin')
    out = str(tmp_path / 'gathered.npy')
    script = tmp_path / 'worker.py'
    script.write_text(WORKER.format(repo=REPO, model=model, out=out, use_hip=use_hip, count=count,
                                    force_host_group=force_host_group))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
    result = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node={}'.format(ranks),
         '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)],
        env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert result.returncode == 0, result.stdout[-3000:]
    gathered = np.load(out)
    assert gathered.shape == (count, 300)
    assert (gathered[0] == 0).all()  # entry 0 is a missing word


def test_two_rank_shard_and_host_gather(native, tmp_path):
    run_two_ranks(tmp_path, use_hip=False, count=1001, port=29571)


def test_eight_rank_shard_and_gather_over_a_separate_gloo_group(native, tmp_path):
    # the N = 8 split the driver's scaling run uses (host path, CPU only), gathered through a gloo group made
    # for the purpose -- what gather_rows does when the job's default group is nccl: the data never enters RCCL
    run_two_ranks(tmp_path, use_hip=False, count=1003, port=29577, ranks=8, force_host_group=True)


SUBGROUP_WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, {repo!r})
    import memb_amd.sharding as sharding

    dist.init_process_group('gloo')
    rank = dist.get_rank()
    sub = dist.new_group(ranks=[0, 2])          # (collective over the whole job, as torch requires)
    if rank in (0, 2):
        # a host group for the sub-group's ranks only: rank 1 neither calls nor is waited for
        hosts = sharding.host_group(sub, force_new=True)
        assert dist.get_backend(hosts) == 'gloo' and dist.get_process_group_ranks(hosts) == [0, 2]
        assert sharding.host_group(sub, force_new=False) is sub          # already gloo: used as it is
        assert sharding._host_groups[(0, 2)] is hosts                     # remembered by its ranks, not by id()
        sharding.host_group = lambda group=None, force_new=False: hosts   # what an nccl job's gather would use
        me = dist.get_rank(sub)
        start, stop = sharding.shard_range(11, me, 2)
        local = np.arange(start, stop, dtype=np.float32)[:, None] * np.ones((1, 3), dtype=np.float32)
        full = sharding.gather_rows(local, 11, group=sub, dst=1)          # dst = rank 1 of the sub-group = global 2
        if rank == 2:
            assert np.array_equal(full[:, 0], np.arange(11, dtype=np.float32))
            open({out!r}, 'w').write('ok')
        else:
            assert full is None
    dist.barrier()
    dist.destroy_process_group()
''')


def test_host_group_of_a_sub_group_is_created_by_its_ranks_only(native, tmp_path):
    out = str(tmp_path / 'sub.txt')
    script = tmp_path / 'sub_worker.py'
    script.write_text(SUBGROUP_WORKER.format(repo=REPO, out=out))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
    result = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=3',
         '--master-addr', '127.0.0.1', '--master-port', '29579', str(script)],
        env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert result.returncode == 0, result.stdout[-3000:]
    assert open(out).read() == 'ok'


@pytest.mark.gpu
def test_two_rank_shard_through_the_hip_path(native, tmp_path):
    # two processes, each decoding its slice on the GPU (above the 512-word small-batch path and below it)
    run_two_ranks(tmp_path, use_hip=True, count=3001, port=29573)
    run_two_ranks(tmp_path, use_hip=True, count=301, port=29575)
