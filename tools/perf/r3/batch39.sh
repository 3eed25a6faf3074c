#!/bin/bash
# round 3, batch 39: the kernel's own duration (rocprofv3 kernel trace) on 100 k and 20 k random rows with phases switched
# off -- the burst timing of batch 28 is bounded by the host's submission rate below ~7 us
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
export MEMB_PACKAGE_ROOT=build/measure
for words in 100000 20000; do
for debug in 0 7 3 6 2 1 5 8197; do
out=gpurun_out/phases_${words}_$debug
PH_WORDS=$words PH_DEBUG=$debug rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 tools/perf/phases.py > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
python3 - $out $words $debug <<'PY'
import csv, glob, sys
out, words, debug = sys.argv[1:]
for row in csv.DictReader(open(glob.glob(out + '/*kernel_stats.csv')[0])):
    if 'decode_' in row['Name']:
        print('%s words debug %5s: %-60s calls %s avg %.2f us min %.2f us' % (words, debug, row['Name'][27:80], row['Calls'], float(row['AverageNs']) / 1e3, float(row['MinNs']) / 1e3), flush=True)
PY
done
done
