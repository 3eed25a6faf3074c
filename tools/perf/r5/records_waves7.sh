#!/bin/bash
# Round 5: the nibble-key records pipeline at SEVEN wavefronts per SIMD (output burst 3 + a scalar-register budget of 96:
# build/recb3) against the tree's six (burst 4). Alternating processes, 4-bit and 2-bit models.
set -o pipefail
out=gpurun_out/r5_records_waves7
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=3
cases=70k,85k,100k,114k,120k,130k,rot70k,rot85k,rot100k,rot114k,rot120k,rot130k
for pass in 1 2; do
    for build in new old; do
        root=""; [ $build = new ] && root=build/recb3
        for model in "4 2196017" "2 2196017"; do
            set -- $model
            MEMB_PACKAGE_ROOT=$root AB3= AB3_BITS=$1 AB3_WORDS=$2 AB3_CASES=$cases timeout -k 10 300 python tools/perf/ab3.py > $out/${build}_$1bit_$pass.txt 2>&1 || { tail -20 $out/${build}_$1bit_$pass.txt; exit 1; }
        done
    done
done
python - <<'PY'
import re
def read(f):
    txt=open(f).read().split('--- median')[1]; case=None; rows={}
    for line in txt.splitlines():
        m=re.match(r'case (\S+)',line)
        if m: case=m.group(1)
        m=re.match(r'\s+base\s+([\d.]+)',line)
        if m and case: rows[case]=float(m.group(1))
    return rows
for bits in (4,2):
    t={(b,p):read('gpurun_out/r5_records_waves7/%s_%dbit_%d.txt'%(b,bits,p)) for b in ('new','old') for p in (1,2)}
    print('%d-bit: case, seven per SIMD pass 1/2, six per SIMD pass 1/2, mean change'%bits)
    for c in t[('new',1)]:
        a=[t[('new',p)][c] for p in (1,2)]; b=[t[('old',p)][c] for p in (1,2)]
        print('  %-8s %.4f %.4f | %.4f %.4f | %+.1f %%'%(c,a[0],a[1],b[0],b[1],100*(sum(a)/sum(b)-1)))
PY
