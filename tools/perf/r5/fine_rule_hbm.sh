#!/bin/bash
# Round 5: the finer-index rule with NOTHING cached between launches (ab3 'hbm<N>k': as many batches and output buffers in
# rotation as it takes to put 640 MB of output between two uses of one): rule / never / always / the records pipeline.
set -o pipefail
out=gpurun_out/r5_fine_rule_hbm
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=3
cases=hbm1k,hbm10k,hbm20k,hbm28k,hbm30k,hbm40k,hbm50k,hbm57k,hbm58k,hbm62k,hbm65k,hbm66k,hbm80k
for model in "4 2196017 1234" "6 1999995 1234"; do
    set -- $model
    AB3='off:fine_lanes=1,on:fine_lanes=2,rec:persistent=2' AB3_BITS=$1 AB3_WORDS=$2 AB3_SEED=$3 AB3_CASES=$cases \
        timeout -k 10 500 python tools/perf/ab3.py > $out/fine_$1bit_$3.txt 2>&1 || { tail -30 $out/fine_$1bit_$3.txt; exit 1; }
done
python - <<'PY'
import re,glob
for f in sorted(glob.glob('gpurun_out/r5_fine_rule_hbm/fine_*.txt')):
    txt=open(f).read().split('--- median')[1]
    case=None; rows={}
    for line in txt.splitlines():
        m=re.match(r'case (\S+)',line)
        if m: case=m.group(1); rows[case]={}
        m=re.match(r'\s+(\w+)\s+([\d.]+)\s+([+-][\d.]+) %',line)
        if m and case: rows[case][m.group(1)]=float(m.group(2))
    print(f)
    for c,r in rows.items(): print('%7s rule %.4f never %.4f always %.4f records %.4f | always/never %+.1f %%  records/never %+.1f %%  rule/best %+.1f %%'%(c,r['base'],r['off'],r['on'],r['rec'],100*(r['on']/r['off']-1),100*(r['rec']/r['off']-1),100*(r['base']/min(r['on'],r['off'],r['rec'])-1)))
PY
