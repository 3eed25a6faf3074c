#!/bin/bash
# Round 5, batch 5: nibble-key models through the 4-byte table entries in the SINGLE-model kernels too (build/v1:
# -DMEMB_HIP_FAST_COMPACT) against the tree (build/base), two builds alternating on one box.
set -o pipefail
out=gpurun_out/r5_batch5
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=3
MEMB_PACKAGE_ROOT=build/v1 timeout -k 10 300 python - > $out/parity.txt 2>&1 <<'PY' || { tail -20 $out/parity.txt; exit 1; }
import os, sys
sys.path.insert(0, os.path.abspath('build/v1'))
sys.path.insert(1, os.getcwd())
import numpy as np, torch, memb_amd, oracle
from memb_amd import synthetic
assert 'build/v1' in memb_amd.__file__
for bits in (4, 2):
    path, _ = synthetic.cached_model(50000, 300, 'trained', bits)
    reader = memb_amd.Reader(path); checker = oracle.OracleReader(path)
    rng = np.random.default_rng(1)
    for count in (1, 1000, 30000, 50000, 200000):
        rows = rng.integers(0, 50000, size=count).astype(np.uint32); rows[::97] = 0xFFFFFFFF
        out = reader.rows_embedding_device(torch.from_numpy(rows.view(np.int32)).cuda())
        assert np.array_equal(out.cpu().numpy().view(np.uint32), checker.rows_embedding(rows).view(np.uint32)), (bits, count)
        assert np.array_equal(reader.rows_embedding(rows).view(np.uint32), checker.rows_embedding(rows).view(np.uint32)), (bits, count)
print('parity ok')
PY
cat $out/parity.txt | tail -1
for round in 1 2; do
    for root in base v1; do
        MEMB_PACKAGE_ROOT=build/$root AB3_CASES=sorted,random,500k,100k,rot100k,10k,1k \
            timeout -k 10 300 python tools/perf/ab3.py > $out/4bit_${root}_$round.txt 2>&1 || { tail -20 $out/4bit_${root}_$round.txt; exit 1; }
        echo "round $round build/$root 4-bit"; sed -n '/--- median/,$p' $out/4bit_${root}_$round.txt | grep "case\|base " | paste - - | awk '{print $2, $4}' | tr '\n' ' '; echo
    done
done
for root in base v1 base v1; do
    MEMB_PACKAGE_ROOT=build/$root AB3_BITS=2 AB3_CASES=sorted,random,100k,10k \
        timeout -k 10 300 python tools/perf/ab3.py > $out/2bit_${root}.txt 2>&1 || exit 1
    echo "build/$root 2-bit"; sed -n '/--- median/,$p' $out/2bit_${root}.txt | grep "case\|base " | paste - - | awk '{print $2, $4}' | tr '\n' ' '; echo
done
