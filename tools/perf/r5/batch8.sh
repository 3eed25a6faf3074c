#!/bin/bash
# Round 5, batch 8: the host API (words in, numpy out) with the word search on the device (memb_hip_decode_words):
# tests, then reader[words] timings at 10 k / 100 k / 2.2 M words against host search + rows -> numpy.
set -o pipefail
out=gpurun_out/r5_batch8
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
timeout -k 10 900 python -m pytest tests/test_gpu_words.py tests/test_cpp_interface.py tests/test_gpu_full_size.py -x -q -k "not union_of_two" > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
timeout -k 10 600 python - > $out/e2e.txt 2>&1 <<'PY' || { tail -20 $out/e2e.txt; exit 1; }
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import memb_amd
from memb_amd import synthetic
n = 2196017
path, _ = synthetic.cached_model(n, 300, 'trained', 4)
r = memb_amd.Reader(path); keys = r.keys(); r.info(); r.stage_words()
rng = np.random.default_rng(3)
for m in (1000, 10000, 100000, n):
    words = keys if m == n else [keys[i] for i in rng.integers(0, n, size=m)]
    reused = np.zeros((m, 300), dtype=np.float32)
    best = [1e9] * 4
    for rep in range(4):
        t0 = time.perf_counter(); rows = r.resolve_rows(words); t1 = time.perf_counter(); r.rows_embedding_into(rows, reused); t2 = time.perf_counter()
        full = r.batch_embedding(words); t3 = time.perf_counter(); del full
        t4 = time.perf_counter(); r.batch_embedding_into(words, reused); t5 = time.perf_counter()
        best = [min(a, b) for a, b in zip(best, (t1 - t0, t2 - t1, t3 - t2, t5 - t4))]
    print('n=%8d host search %.3f ms | rows -> reused numpy %.3f ms | reader[words] (fresh result) %.3f ms = %.1f M emb/s | into a reused result %.3f ms = %.1f M emb/s' % (
        m, best[0] * 1e3, best[1] * 1e3, best[2] * 1e3, m / best[2] / 1e6, best[3] * 1e3, m / best[3] / 1e6), flush=True)
PY
cat $out/e2e.txt
