"""Batch sharding across GPUs, one process per GPU.

Every word decodes from its own byte-aligned stream and read-only tables, so a
batch splits into independent slices with no exchange between them: rank g
takes entries [g * ceil(n / G), (g + 1) * ceil(n / G)) -- the same split the
reference uses for its host threads (src/reader.cpp:65-79) -- and every rank
holds a full replica of the model (<= 0.4 GB, nothing next to 288 GB of HBM).
There is no collective on the data path. `gather_rows` is the optional
host-side gather for callers that want the whole matrix in one place.
"""
import numpy as np


def shard_range(count, rank, world_size):
    '''[start, stop) of the batch entries rank `rank` of `world_size` looks up'''
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError('rank {} outside world of {}'.format(rank, world_size))
    job_size = (count + world_size - 1) // world_size
    start = min(count, rank * job_size)
    return start, min(count, start + job_size)


def shard_of(items, rank, world_size):
    start, stop = shard_range(len(items), rank, world_size)
    return items[start:stop]


def lookup_shard(reader, words, rank, world_size):
    '''This rank's rows of reader.batch_embedding(words): (stop - start, dim) float32'''
    return reader.batch_embedding(list(shard_of(words, rank, world_size)))


def gather_rows(local_rows, count, group=None, dst=0):
    '''Host-side gather of the per-rank slices into one (count, dim) array on rank `dst`
    (None elsewhere). Uses the process group only to move host buffers.'''
    import torch
    import torch.distributed as dist

    world_size = dist.get_world_size(group)
    rank = dist.get_rank(group)
    local_rows = np.ascontiguousarray(local_rows, dtype=np.float32)
    dim = local_rows.shape[1]
    job_size = (count + world_size - 1) // world_size
    device = 'cuda' if dist.get_backend(group) == 'nccl' else 'cpu'
    padded = torch.zeros((job_size, dim), dtype=torch.float32, device=device)
    padded[:local_rows.shape[0]] = torch.from_numpy(local_rows).to(device)
    pieces = [torch.empty_like(padded) for _ in range(world_size)] if rank == dst else None
    dist.gather(padded, pieces, dst=dst, group=group)
    if rank != dst:
        return None
    out = np.empty((count, dim), dtype=np.float32)
    for r, piece in enumerate(pieces):
        start, stop = shard_range(count, r, world_size)
        out[start:stop] = piece[:stop - start].cpu().numpy()
    return out
