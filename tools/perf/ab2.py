"""Interleaved A/B of kernel variants inside one process.

    AB2='base:0,sc1:32,w4:0:MEMB_HIP_WAVES=4' python tools/perf/ab2.py

Each variant = name:MEMB_HIP_DEBUG[:ENV=VALUE;ENV=VALUE...]; one Reader per variant (the switches
are read when a Reader's device side is created), `AB2_ROUNDS` interleaved rounds, full dump in key
order / the same rows shuffled / 100 000 random rows. AB2_BITS, AB2_WORDS choose the model.
"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

import memb_amd
from memb_amd import synthetic

n = int(os.environ.get('AB2_WORDS', '2196017'))
bits = int(os.environ.get('AB2_BITS', '4'))
rounds = int(os.environ.get('AB2_ROUNDS', '3'))
reps = int(os.environ.get('AB2_REPS', '20'))
cases = os.environ.get('AB2_CASES', 'sorted,random,100k').split(',')
path, _ = synthetic.cached_model(n, 300, 'trained', bits)
out = torch.empty((n, 300), dtype=torch.float32, device='cuda')
rows = torch.arange(n, dtype=torch.int32, device='cuda')
perm = torch.randperm(n, device='cuda').to(torch.int32)
small = perm[:100000].contiguous()
small_out = torch.empty((100000, 300), dtype=torch.float32, device='cuda')


flush = torch.empty(1 << 28, dtype=torch.float32, device='cuda') if any(c.startswith('cold') for c in cases) else None


def timeit(f, cold=False):
    """cold: a 1 GiB fill between launches, so nothing of the model is left in L2 / the Infinity Cache"""
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    f()
    b.record()
    torch.cuda.synchronize()
    one = max(a.elapsed_time(b), 1e-3)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    # ~20 ms of the same launches without a gap first: the part's power state settles (tools/perf/ramp.py)
    for _ in range(max(3, min(2000, int(float(os.environ.get('AB2_RUN_IN_MS', '20')) / one) + 1))):
        f()
    for a, b in ev:
        if cold:
            flush.fill_(1.0)
        a.record()
        f()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[0], ms[len(ms) // 2]


variants = []
for spec in os.environ.get('AB2', 'base:0').split(','):
    parts = spec.split(':')
    env = dict(kv.split('=') for kv in parts[2].split(';')) if len(parts) > 2 and parts[2] else {}
    variants.append((parts[0], parts[1], env))

readers = {}
for name, flags, env in variants:
    saved = {k: os.environ.get(k) for k in list(env) + ['MEMB_HIP_DEBUG']}
    os.environ['MEMB_HIP_DEBUG'] = flags
    os.environ.update(env)
    readers[name] = memb_amd.Reader(path, device=0)
    info = readers[name].info()
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    print('variant %-10s flags %-5s env %s -> waves/block %d lds %d' % (name, flags, env, info['waves_per_block'], info['lds_bytes_per_block']), flush=True)

# every variant must produce the same bits as the first
reference = None
for name, _, _ in variants:
    readers[name].rows_embedding_device(small, out=small_out)
    torch.cuda.synchronize()
    got = small_out.clone()
    if reference is None:
        reference = got
    elif not torch.equal(reference.view(torch.int32), got.view(torch.int32)):
        print('variant %s: OUTPUT DIFFERS from %s' % (name, variants[0][0]), flush=True)

best = {}
for rnd in range(rounds):
    for name, _, _ in variants:
        r = readers[name]
        line = 'round %d %-10s' % (rnd, name)
        for case in cases:
            if case == 'sorted':
                t = timeit(lambda: r.rows_embedding_device(rows, out=out))
            elif case == 'coldsorted':
                t = timeit(lambda: r.rows_embedding_device(rows, out=out), cold=True)
            elif case == 'coldrandom':
                t = timeit(lambda: r.rows_embedding_device(perm, out=out), cold=True)
            elif case == 'random':
                t = timeit(lambda: r.rows_embedding_device(perm, out=out))
            else:
                t = timeit(lambda: r.rows_embedding_device(small, out=small_out))
            best[(name, case)] = min(best.get((name, case), 1e9), t[1])
            line += ' | %s min %.4f med %.4f' % (case, t[0], t[1])
        print(line, flush=True)
print('--- best medians (ms)')
for name, _, _ in variants:
    print('%-10s ' % name + '  '.join('%s %.4f' % (case, best[(name, case)]) for case in cases), flush=True)
