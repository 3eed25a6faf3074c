#!/bin/bash
# HBM traffic of the bench kernel: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots)
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_$tag
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_$tag/fetch -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-live-traffic > /dev/null 2> gpurun_out/prof_$tag/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_$tag/write -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-live-traffic > /dev/null 2> gpurun_out/prof_$tag/write.err
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d gpurun_out/prof_$tag/tcc -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-live-traffic > /dev/null 2> gpurun_out/prof_$tag/tcc.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d gpurun_out/prof_$tag/tcc2 -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-live-traffic > /dev/null 2> gpurun_out/prof_$tag/tcc2.err
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/prof_$tag/*/*counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r['Kernel_Name'][:60], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print('%-62s %-24s n=%d mean=%.5g' % (k[0], k[1], len(v), sum(v)/len(v)))
PY
tail -2 gpurun_out/prof_$tag/tcc.err
