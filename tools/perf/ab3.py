"""Interleaved A/B of kernel variants on ONE Reader (one allocation), with an A/A control.

    AB3='t2:tiles_per_wave=2,w8:waves_per_block=8' python tools/perf/ab3.py

Why: tools/perf/ab2.py gave every variant a Reader -- and so a 351 MB stream array -- of its own, and two
variants that were IDENTICAL machine code differed by 2-4 % in every round: it measured where
the allocator had put each copy. Here
  * every variant is a set of run-time options (memb_hip_ctx_set_option; `debug=N` needs the measurement
    build: tools/perf/build_measure.py, MEMB_PACKAGE_ROOT=build/measure) applied to the SAME context;
  * the baseline is listed twice ('base' and 'base2'): their difference is the floor below which no
    decision stands; rounds alternate between the given order and its reverse;
  * a variant that cannot be an option (a layout chosen when the model is staged) is written
    `name:!ENV=VALUE;...` and gets a Reader of its own -- flagged '(own allocation)' in the table;
  * AB3_PLACEMENT=N adds N more Readers with the baseline's settings: same code, different allocations;
    AB3_OUT_BUFFERS=K times the baseline into K different output buffers.
Cases (AB3_CASES): sorted = full dump in key order, random = the same rows shuffled, 100k / 10k / 1k =
random batches; cold* = with a 1 GiB fill between launches. AB3_BITS / AB3_WORDS choose the model.
"""
import os
import sys
import time

ROOT = os.environ.get('MEMB_PACKAGE_ROOT') or os.getcwd()
sys.path.insert(0, os.path.abspath(ROOT))
import numpy as np
import torch

import memb_amd
from memb_amd import synthetic

n = int(os.environ.get('AB3_WORDS', '2196017'))
bits = int(os.environ.get('AB3_BITS', '4'))
rounds = int(os.environ.get('AB3_ROUNDS', '4'))
reps = int(os.environ.get('AB3_REPS', '20'))
run_in_ms = float(os.environ.get('AB3_RUN_IN_MS', '20'))
cases = os.environ.get('AB3_CASES', 'sorted,random,100k').split(',')
placement = int(os.environ.get('AB3_PLACEMENT', '0'))
out_buffers = int(os.environ.get('AB3_OUT_BUFFERS', '0'))
DEFAULTS = {'fine_lanes': 0, 'union_split': 1, 'tiles_per_wave': 0, 'waves_per_block': 0, 'persistent': 1}

print('package: %s   model: %d words, %d-bit seed %s %s   rounds %d x %d launches after %.0f ms run-in' % (
    os.path.dirname(memb_amd.__file__), n, bits, os.environ.get('AB3_SEED', '1234'), os.environ.get('AB3_DIST', 'normal'), rounds, reps, run_in_ms), flush=True)
seed = int(os.environ.get('AB3_SEED', '1234'))                 # 99: a 4-bit model with a 9-bit code (byte keys)
distribution = os.environ.get('AB3_DIST', 'normal')           # or 'student'
path, _ = synthetic.cached_model(n, 300, 'trained', bits, seed=seed, distribution=distribution)
out = torch.empty((n, 300), dtype=torch.float32, device='cuda')
rows = torch.arange(n, dtype=torch.int32, device='cuda')
generator = torch.Generator(device='cuda')
generator.manual_seed(5)
perm = torch.randperm(n, device='cuda', generator=generator).to(torch.int32)
batches = {'100k': perm[:100000].contiguous(), '10k': perm[100000:110000].contiguous(), '1k': perm[110000:111000].contiguous(),
           '500k': perm[200000:700000].contiguous(), '250k': perm[700000:950000].contiguous(), '50k': perm[950000:1000000].contiguous(),
           '5k': perm[1000000:1005000].contiguous(), '20k': perm[1010000:1030000].contiguous(), '30k': perm[1030000:1060000].contiguous(),
           '16k': perm[1060000:1076384].contiguous(), '40k': perm[1080000:1120000].contiguous()}
flush = torch.empty(1 << 28, dtype=torch.float32, device='cuda') if any(c.startswith('cold') for c in cases) else None


def timeit(call, cold=False):
    call()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    call()
    b.record()
    torch.cuda.synchronize()
    one = max(a.elapsed_time(b), 1e-3)
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for _ in range(max(3, min(4000, int(run_in_ms / one) + 1))):   # the part's power state settles (tools/perf/ramp.py)
        call()
    if one < 0.2 and not cold:
        # an event pair per launch adds 4-5 us: short kernels as the average of a burst between ONE pair of events
        a, b = events[0]
        a.record()
        for _ in range(5 * reps):
            call()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / (5 * reps)
    for a, b in events:
        if cold:
            flush.fill_(1.0)
        a.record()
        call()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in events)
    return ms[len(ms) // 2]


def parse(spec):
    name, _, rest = spec.partition(':')
    options, env = {}, {}
    for item in filter(None, rest.split(';')):
        key, _, value = item.partition('=')
        if key.startswith('!') or key.startswith('MEMB_'):
            env[key.lstrip('!')] = value
        else:
            options[key] = int(value, 0)
    return name, options, env


def open_reader(env):
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    reader = memb_amd.Reader(path, device=0)
    reader.info()
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    return reader


shared = open_reader({})
variants = [('base', {}, {}, shared)]
for spec in filter(None, os.environ.get('AB3', '').split(',')):
    name, options, env = parse(spec)
    variants.append((name, options, env, open_reader(env) if env else shared))
variants.append(('base2', {}, {}, shared))
for index in range(placement):
    variants.append(('place%d' % index, {}, {'MEMB_HIP_PLACEMENT_PROBE': str(index)}, open_reader({})))
outputs = [out]
for index in range(out_buffers):
    outputs.append(torch.empty((n, 300), dtype=torch.float32, device='cuda'))


def apply(reader, options):
    settings = dict(DEFAULTS)
    if 'debug' in options or 'lds_pad' in options or os.environ.get('MEMB_PACKAGE_ROOT'):
        settings['debug'] = 0
        settings['lds_pad'] = 0   # (round 5's first residency table ran without this line: a variant kept the pad of the one before it)
    settings.update(options)
    for key, value in settings.items():
        try:
            reader.set_option(key, value)
        except RuntimeError:
            if key not in ('debug', 'lds_pad') or value:   # (a package root that is not a measurement build has no such option)
                raise


for name, options, env, reader in variants:
    apply(reader, options)
    info = reader.info()
    print('variant %-10s %s%s -> %s, waves/block %d, lds %d%s' % (
        name, options, ' env %s' % env if env else '', info['kernel'], info['waves_per_block'], info['lds_bytes_per_block'],
        '' if reader is shared else '  (own allocation)'), flush=True)

# every variant that claims to produce results must produce the baseline's bits
small_out = torch.empty((100000, 300), dtype=torch.float32, device='cuda')
burst_mode = os.environ.get('AB3_BURST', '0') == '1'   # short kernels: average of a burst between one event pair
reference = None
for name, options, env, reader in variants:
    apply(reader, options)
    reader.rows_embedding_device(batches['100k'], out=small_out)
    torch.cuda.synchronize()
    got = small_out.clone()
    if reference is None:
        reference = got
    elif not torch.equal(reference.view(torch.int32), got.view(torch.int32)):
        print('variant %s: output differs from base (expected for measurement switches that skip work)' % name, flush=True)


union = None
if any(c.endswith('union') for c in cases):
    # BASELINE.json configs[4]: concatenation of two 4-bit models, 500 000 words, a quarter missing per model;
    # the options of the FIRST reader choose the kernel
    from memb_amd import _memb
    second_path, _ = synthetic.cached_model(1999995, 300, 'trained', int(os.environ.get('AB3_UNION_BITS', '4')), seed=4321)
    second = memb_amd.Reader(second_path, device=0)
    second.info()
    rng = np.random.default_rng(17)
    batch = int(os.environ.get('AB3_UNION_WORDS', '500000'))
    ids = []
    for count in (n, len(second)):
        picks = rng.integers(0, count, size=batch).astype(np.uint32)
        picks[rng.random(batch) < 0.25] = 0xFFFFFFFF
        ids.append(torch.from_numpy(picks.view(np.int32)).cuda())
    merged = torch.empty((batch, 600), dtype=torch.float32, device='cuda')
    union = (second, ids, merged, batch)
    # 'hbmunion': other ids and another output buffer every launch, 640 MB of output between two uses of one
    union_sets = []
    if 'hbmunion' in cases:
        for k in range(max(4, -(-640000000 // (batch * 2400)))):
            more = []
            for count in (n, len(second)):
                picks = rng.integers(0, count, size=batch).astype(np.uint32)
                picks[rng.random(batch) < 0.25] = 0xFFFFFFFF
                more.append(torch.from_numpy(picks.view(np.int32)).cuda())
            union_sets.append((more, torch.empty((batch, 600), dtype=torch.float32, device='cuda')))
    union_turn = [0]


def run_union(reader, rotate=False):
    second, ids, merged, batch = union
    if rotate:
        ids, merged = union_sets[union_turn[0] % len(union_sets)]
        union_turn[0] += 1
    done = _memb.union_rows_to_device(
        [reader._impl, second._impl], [ids[0].data_ptr(), ids[1].data_ptr()], [0, 300], batch, merged.data_ptr(),
        merged.stride(0), torch.cuda.current_stream().cuda_stream, False)
    assert done


def run_case(reader, case, target):
    cold = case.startswith('cold')
    kind = case[4:] if cold else case
    if kind == 'union':
        return timeit(lambda: run_union(reader), cold)
    if kind == 'hbmunion':
        return timeit(lambda: run_union(reader, True), cold)
    if kind == 'sorted':
        return timeit(lambda: reader.rows_embedding_device(rows, out=target), cold)
    if kind == 'random':
        return timeit(lambda: reader.rows_embedding_device(perm, out=target), cold)
    if kind.startswith('rot') or kind.startswith('hbm'):
        # 'rot<N>k': FOUR different batches of N thousand random rows round-robin into four output buffers -- more than the
        # 256 MB Infinity Cache holds from one turn to the next at 100 k rows: every launch reads from and writes to HBM
        # 'hbm<N>k': the same with as many batches as it takes to put 640 MB of output between two uses of a buffer -- the
        # rotating regime for batches too small for four of them to outgrow the cache
        count = int(kind[3:-1]) * 1000
        if kind not in rotating:
            sets = []
            how_many = 4 if kind.startswith('rot') else max(4, -(-640000000 // (count * 1200)))
            for k in range(how_many):
                ids = perm[(300000 + k * count) % (n - count):][:count].contiguous()
                sets.append((ids, torch.empty((count, 300), dtype=torch.float32, device='cuda')))
            rotating[kind] = (sets, [0])
        sets, turn = rotating[kind]

        def call():
            ids, target_k = sets[turn[0] % len(sets)]
            turn[0] += 1
            reader.rows_embedding_device(ids, out=target_k)
        return timeit(call, cold)
    if kind not in batches:   # any '<N>k': N thousand random rows
        count = int(kind[:-1]) * 1000
        batches[kind] = perm[1000000:1000000 + count].contiguous()
    batch = batches[kind]
    view = target[:len(batch)]
    return timeit(lambda: reader.rows_embedding_device(batch, out=view), cold)


rotating = {}
results = {}   # (variant, case) -> [median of each round]
started = time.time()
for rnd in range(rounds):
    order = variants if rnd % 2 == 0 else list(reversed(variants))
    for name, options, env, reader in order:
        apply(reader, options)
        line = 'round %d %-10s' % (rnd, name)
        for case in cases:
            median = run_case(reader, case, out)
            results.setdefault((name, case), []).append(median)
            line += ' | %s %.4f' % (case, median)
        print(line, flush=True)
if out_buffers:
    apply(shared, {})
    for index, target in enumerate(outputs):
        line = 'output buffer %d @%#x' % (index, target.data_ptr())
        for case in cases:
            values = [run_case(shared, case, target) for _ in range(rounds)]
            results[('out%d' % index, case)] = values
            line += ' | %s %s' % (case, ' '.join('%.4f' % v for v in values))
        print(line, flush=True)

print('--- median over rounds (ms); delta vs base; [min .. max] over rounds   (%.0f s)' % (time.time() - started))
names = [v[0] for v in variants] + (['out%d' % i for i in range(len(outputs))] if out_buffers else [])
for case in cases:
    base = sorted(results[('base', case)])
    base_median = base[len(base) // 2]
    print('case %s' % case)
    for name in names:
        values = sorted(results[(name, case)])
        median = values[len(values) // 2]
        print('  %-10s %.4f  %+6.2f %%   [%.4f .. %.4f]' % (name, median, 100.0 * (median / base_median - 1.0), values[0], values[-1]))
    pair = [abs(a / b - 1.0) for a, b in zip(results[('base', case)], results[('base2', case)])]
    both = sorted(results[('base', case)] + results[('base2', case)])
    print('  A/A floor: base vs base2 per round max %.2f %%, medians %.2f %%, all %d base timings span %.2f %%' % (
        100.0 * max(pair), 100.0 * abs(sorted(results[('base2', case)])[len(base) // 2] / base_median - 1.0),
        len(both), 100.0 * (both[-1] / both[0] - 1.0)))
