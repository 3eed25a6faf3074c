#!/bin/bash
# Round 5: the edges of decode_records_persistent's class again (2 R < tiles <= 4 R), now that it runs six wavefronts per SIMD:
# rule (base) / one tile per wavefront always (one) / the pipeline wherever the layout allows (rec), outside the class.
set -o pipefail
out=gpurun_out/r5_records_edges
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=3
cases=50k,60k,66k,135k,150k,180k,250k,500k,rot50k,rot60k,rot66k,rot135k,rot150k,rot180k,random
for model in "4 2196017" "6 1999995"; do
    set -- $model
    AB3='one:persistent=0,rec:persistent=2' AB3_BITS=$1 AB3_WORDS=$2 AB3_CASES=$cases \
        timeout -k 10 500 python tools/perf/ab3.py > $out/$1bit.txt 2>&1 || { tail -30 $out/$1bit.txt; exit 1; }
done
python - <<'PY'
import re,glob
for f in sorted(glob.glob('gpurun_out/r5_records_edges/*bit.txt')):
    txt=open(f).read().split('--- median')[1]; case=None; rows={}
    for line in txt.splitlines():
        m=re.match(r'case (\S+)',line)
        if m: case=m.group(1); rows[case]={}
        m=re.match(r'\s+(\w+)\s+([\d.]+)\s+([+-][\d.]+) %',line)
        if m and case: rows[case][m.group(1)]=float(m.group(2))
    print(f)
    for c,r in rows.items(): print('  %8s rule %.4f one-tile %.4f records %.4f   rec/one %+.1f %%   rule/best %+.1f %%'%(c,r['base'],r['one'],r['rec'],100*(r['rec']/r['one']-1),100*(r['base']/min(r['one'],r['rec'])-1)))
PY
