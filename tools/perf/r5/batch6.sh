#!/bin/bash
# Round 5, batch 6: the loader / storer pattern experiment, and the word search under rocprofv3.
set -o pipefail
out=gpurun_out/r5_batch6
mkdir -p $out
export MEMB_SYNTH_DEVICE=0
timeout -k 10 600 python tools/perf/r5/specialised.py > $out/specialised.txt 2>&1 || { tail -30 $out/specialised.txt; exit 1; }
cat $out/specialised.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/resolve_trace -o trace -- python3 tools/perf/r5/words.py --repeats 5 > $out/resolve_words.txt 2> $out/resolve_trace.err || { tail -20 $out/resolve_trace.err; exit 1; }
cp $out/resolve_trace/*kernel_stats.csv $out/r05_resolve_kernel_stats.csv
grep -i "resolve_words\|build_word_table\|Name" $out/r05_resolve_kernel_stats.csv
grep "all\|random\|stage" $out/resolve_words.txt
