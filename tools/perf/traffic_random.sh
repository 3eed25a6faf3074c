#!/bin/bash
# read traffic of a random 100k-row batch (configs[1]): how many bytes per word does a random lookup fetch?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_r2_random; mkdir -p $out
args="bench.py --workload glove840b-300d-4bit-100k --no-configs --no-cpu-baseline --no-live-traffic --steps 5 --warmup 2"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o pmc -- python3 $args > /dev/null 2> $out/fetch.err
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $out/tcc -o pmc -- python3 $args > /dev/null 2> $out/tcc.err
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum --output-format csv -d $out/tcc2 -o pmc -- python3 $args > /dev/null 2> $out/tcc2.err
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob('$out/*/*counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'decode_trained_persistent' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print('%-24s n=%d mean=%.5g  per word %.1f' % (k, len(v), sum(v)/len(v), sum(v)/len(v)/100000))
PY
