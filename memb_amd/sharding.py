"""Batch sharding across GPUs, one process per GPU.

Every word decodes from its own byte-aligned stream and read-only tables, so a
batch splits into independent slices with no exchange between them: rank g
takes entries [g * ceil(n / G), (g + 1) * ceil(n / G)) -- the same split the
reference uses for its host threads (src/reader.cpp:65-79) -- and every rank
holds a full replica of the model (<= 0.4 GB, nothing next to 288 GB of HBM).
There is no collective on the data path. `gather_rows` is the optional
host-side gather for callers that want the whole matrix in one place.
"""
import threading

import numpy as np

from .reader import BaseReader, Reader, tokenizer_word_list


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


_host_groups = {}


def host_group(group=None, force_new=False):
    '''A gloo (host memory, TCP / shared memory) group with the ranks of `group`: the lookup path has no
    exchange step, so nothing of it may enter RCCL -- also when the job's default group is `nccl`.
    Creating the group is collective over the ranks of `group` only (`use_local_synchronization`; ranks
    outside a sub-group neither call nor wait): all of them must reach their first gather_rows call, or
    call host_group(group) themselves at a point they all pass. Groups are remembered by their ranks and by the
    default group they were made under: after destroy_process_group / init_process_group the remembered ones are of a
    world that no longer exists and are dropped; a group replaced with force_new is destroyed (its sockets with it).'''
    import torch.distributed as dist
    if dist.get_backend(group) == 'gloo' and not force_new:
        return group
    world = dist.group.WORLD
    if _host_groups.get('world') is not world:
        _host_groups.clear()
        _host_groups['world'] = world
    ranks = tuple(dist.get_process_group_ranks(group)) if group is not None else None
    key = ranks if ranks is not None and len(ranks) != dist.get_world_size() else None
    if force_new and key in _host_groups:
        try:
            dist.destroy_process_group(_host_groups.pop(key))
        except Exception:   # (a group whose backend is already gone: nothing left to release)
            pass
    if key not in _host_groups:
        if key is None:
            _host_groups[key] = dist.new_group(backend='gloo')
        else:
            _host_groups[key] = dist.new_group(ranks=list(ranks), backend='gloo', use_local_synchronization=True)
    return _host_groups[key]


def gather_rows(local_rows, count, group=None, dst=0):
    '''Host-side gather of the per-rank slices into one (count, dim) array on rank `dst` (a rank of
    `group`; None is returned elsewhere). Host buffers over a gloo group: no device memory, no RCCL.'''
    import torch
    import torch.distributed as dist

    world_size = dist.get_world_size(group)
    rank = dist.get_rank(group)
    local_rows = np.ascontiguousarray(local_rows, dtype=np.float32)
    dim = local_rows.shape[1]
    job_size = (count + world_size - 1) // world_size
    padded = torch.zeros((job_size, dim), dtype=torch.float32)
    padded[:local_rows.shape[0]] = torch.from_numpy(local_rows)
    pieces = [torch.empty_like(padded) for _ in range(world_size)] if rank == dst else None
    hosts = host_group(group)
    global_dst = dist.get_global_rank(group, dst) if group is not None else dst
    dist.gather(padded, pieces, dst=global_dst, group=hosts)
    if rank != dst:
        return None
    out = np.empty((count, dim), dtype=np.float32)
    for r, piece in enumerate(pieces):
        start, stop = shard_range(count, r, world_size)
        out[start:stop] = piece[:stop - start].numpy()
    return out


class ShardedReader(BaseReader):
    '''One model replicated on several GPUs of a node, driven from one process.
    A batch is split into contiguous slices (shard_range), every device looks up and
    decodes its slice and writes it straight into its rows of the one result array
    (small batches: resolved to row ids once on the host first): the "gather" is that the slices are disjoint
    ranges of one host buffer. No collective, no peer traffic.
    Parameters
    ----------
    filename : str or pathlib.Path
    devices : list of int, HIP device indices (a device may be listed more than once)
    num_threads : int, host threads for the word search (0 = all cores)
    '''

    def __init__(self, filename, devices, num_threads=0):
        super().__init__()
        if not devices:
            raise ValueError('at least one device is needed')
        self._readers = [Reader(filename, num_threads, device=device) for device in devices]

    @property
    def dim(self):
        return self._readers[0].dim

    @property
    def devices(self):
        return [reader.device for reader in self._readers]

    def __len__(self):
        return len(self._readers[0])

    def keys(self):
        return self._readers[0].keys()

    def word_embedding(self, word):
        return self._readers[0].word_embedding(word)

    def rows_embedding(self, rows):
        '''Rows by id, slices decoded concurrently on all devices'''
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        out = np.empty((len(rows), self.dim), dtype=np.float32)
        world = len(self._readers)
        errors = []

        def work(rank):
            start, stop = shard_range(len(rows), rank, world)
            try:
                if stop > start:
                    self._readers[rank].rows_embedding_into(rows[start:stop], out[start:stop])
            except Exception as error:  # re-raised on the calling thread
                errors.append(error)

        threads = [threading.Thread(target=work, args=(rank,)) for rank in range(1, world)]
        for thread in threads:
            thread.start()
        work(0)
        for thread in threads:
            thread.join()
        if errors:
            raise errors[0]
        return out

    def batch_embedding(self, words):
        '''Words in, one host matrix out. Batches large enough for every device to search its own slice ON the device
        (memb_amd.Reader does from 4096 words on: memb_hip_decode_words) are cut into word slices -- each device's thread
        packs, searches and decodes its slice --; smaller ones are resolved once on the host and decoded by row id.'''
        words = words if isinstance(words, list) else list(words)
        world = len(self._readers)
        if len(words) < 4096 * world:
            return self.rows_embedding(self._readers[0].resolve_rows(words))
        out = np.empty((len(words), self.dim), dtype=np.float32)
        errors = []

        def work(rank):
            start, stop = shard_range(len(words), rank, world)
            try:
                if stop > start:
                    self._readers[rank].batch_embedding_into(words[start:stop], out[start:stop])
            except Exception as error:  # re-raised on the calling thread
                errors.append(error)

        threads = [threading.Thread(target=work, args=(rank,)) for rank in range(1, world)]
        for thread in threads:
            thread.start()
        work(0)
        for thread in threads:
            thread.join()
        if errors:
            raise errors[0]
        return out

    def tokenizer_embedding(self, tokenizer):
        return self.batch_embedding(tokenizer_word_list(tokenizer))
