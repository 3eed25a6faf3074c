"""The N > 1 path on CPU: two processes (gloo), each takes its slice of the
batch, slices are gathered on the host and must equal the unsharded result.
The per-slice lookup uses the CPU checker here (no GPU in this suite); the
GPU suite runs the same split through the HIP path in one process."""
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
    reader = oracle.OracleReader({model!r}, 1)
    keys = reader.keys()
    batch = [keys[(7 * i) % len(keys)] if i % 9 else 'missing-%d' % i for i in range(1001)]
    mine = shard_of(batch, rank, world)
    local = reader.batch_embedding(mine)
    full = gather_rows(local, len(batch), dst=0)
    if rank == 0:
        assert np.array_equal(full.view(np.uint32), reader.batch_embedding(batch).view(np.uint32))
        np.save({out!r}, full)
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()
''')


def test_two_rank_shard_and_host_gather(native, tmp_path):
    model = os.path.join(GOLDEN, 'synthetic_4bit.bin')
    out = str(tmp_path / 'gathered.npy')
    script = tmp_path / 'worker.py'
    script.write_text(WORKER.format(repo=REPO, model=model, out=out))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
    result = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
         '--master-addr', '127.0.0.1', '--master-port', '29571', str(script)],
        env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert result.returncode == 0, result.stdout[-3000:]
    gathered = np.load(out)
    assert gathered.shape == (1001, 300)
    assert (gathered[0] == 0).all()  # entry 0 is a missing word
