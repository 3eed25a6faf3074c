"""Round 5: device lookups captured into a HIP graph (torch.cuda.CUDAGraph on the capture stream) -- do they capture, do
they replay bit-exact, and what does a replay of K small lookups cost against K eager launches?"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import memb_amd
from memb_amd import synthetic

os.environ.setdefault('MEMB_SYNTH_DEVICE', '0')
path, _ = synthetic.cached_model(2196017, 300, 'trained', 4)
reader = memb_amd.Reader(path)
count = len(reader)
rng = np.random.default_rng(5)


def timed(fn, reps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    start = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - start) / reps * 1e6


for n, k in ((1000, 8), (1000, 16), (10000, 8), (1000, 1), (100000, 4)):
    ids = [torch.from_numpy(rng.integers(0, count, size=n).astype(np.int32)).cuda() for _ in range(k)]
    outs = [torch.empty((n, 300), dtype=torch.float32, device='cuda') for _ in range(k)]
    eager = [torch.empty((n, 300), dtype=torch.float32, device='cuda') for _ in range(k)]

    def run(targets):
        for i in range(k):
            reader.rows_embedding_device(ids[i], out=targets[i])

    def run_many(targets):
        reader.rows_embedding_device_many([(ids[i], targets[i]) for i in range(k)])

    run(eager)                      # warm-up outside the capture (first use raises the kernels' LDS limit)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        run(outs)
    torch.cuda.synchronize()
    for o in outs:
        o.zero_()
    with torch.cuda.graph(graph, stream=side):
        run(outs)
    graph.replay()
    torch.cuda.synchronize()
    same = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(outs, eager))
    # new ids in the same buffers: a replay reads them
    fresh = [torch.from_numpy(rng.integers(0, count, size=n).astype(np.int32)).cuda() for _ in range(k)]
    for i in range(k):
        ids[i].copy_(fresh[i])
    run(eager)
    graph.replay()
    torch.cuda.synchronize()
    same2 = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(outs, eager))
    # the same K lookups as ONE launch (memb_hip_decode_batches_device), captured as well: no Python between replays
    many_graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        run_many(outs)
    torch.cuda.synchronize()
    for o in outs:
        o.zero_()
    with torch.cuda.graph(many_graph, stream=side):
        run_many(outs)
    many_graph.replay()
    torch.cuda.synchronize()
    same3 = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(outs, eager))
    print('%6d rows x %d: replay == eager %s / after new ids %s / one launch %s | eager %.1f us, one launch for all %.1f us, graph replay %.1f us, graph of the one launch %.1f us' % (
        n, k, same, same2, same3, timed(lambda: run(eager)), timed(lambda: run_many(eager)), timed(graph.replay), timed(many_graph.replay)), flush=True)

# K small lookups in one launch: GPU time by HIP events, with the finer index by rule (0), never (1)
for n, k in ((1000, 8), (1000, 16), (2000, 8), (5000, 8)):
    ids = [torch.from_numpy(rng.integers(0, count, size=n).astype(np.int32)).cuda() for _ in range(k)]
    outs = [torch.empty((n, 300), dtype=torch.float32, device='cuda') for _ in range(k)]
    entries = [(ids[i], outs[i]) for i in range(k)]
    line = []
    for option in (0, 1):
        reader.set_option('fine_lanes', option)
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            reader.rows_embedding_device_many(entries)
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(20):
                reader.rows_embedding_device_many(entries)
        graph.replay(); torch.cuda.synchronize()
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = []
        for _ in range(5):
            start.record(); graph.replay(); stop.record(); torch.cuda.synchronize()
            best.append(start.elapsed_time(stop) * 1000 / 20)
        line.append(sorted(best)[2])
    reader.set_option('fine_lanes', 0)
    print('%d x %d rows in one launch: %.2f us by rule, %.2f us with the usual index' % (k, n, line[0], line[1]), flush=True)
