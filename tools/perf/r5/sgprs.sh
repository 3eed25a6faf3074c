#!/bin/bash
# Round 5: the decode kernels with 100 scalar registers (hardware admits 6 wavefronts per SIMD) against the same kernels
# held to 80 (8 per SIMD). Two builds, alternating processes on one box: in-tree = 80, build/sgpr100 = unlimited.
set -o pipefail
out=gpurun_out/r5_sgprs
mkdir -p $out
export MEMB_SYNTH_DEVICE=0 AB3_ROUNDS=3
cases=sorted,random,1000k,500k,100k,rot100k,55k,50k,30k,10k,1k
for pass in 1 2; do
    for build in new old; do
        root=""; [ $build = old ] && root=build/sgpr100
        for model in "4 2196017" "6 1999995"; do
            set -- $model
            MEMB_PACKAGE_ROOT=$root AB3= AB3_BITS=$1 AB3_WORDS=$2 AB3_CASES=$cases timeout -k 10 300 python tools/perf/ab3.py > $out/${build}_$1bit_$pass.txt 2>&1 || { tail -20 $out/${build}_$1bit_$pass.txt; exit 1; }
        done
    done
done
python - <<'PY'
import re,glob
def read(f):
    txt=open(f).read().split('--- median')[1]; case=None; rows={}
    for line in txt.splitlines():
        m=re.match(r'case (\S+)',line)
        if m: case=m.group(1)
        m=re.match(r'\s+base\s+([\d.]+)',line)
        if m and case: rows[case]=float(m.group(1))
    return rows
for bits in (4,6):
    t={(b,p):read('gpurun_out/r5_sgprs/%s_%dbit_%d.txt'%(b,bits,p)) for b in ('new','old') for p in (1,2)}
    print('%d-bit: case, the tree (MEMB_HIP_SGPRS as shipped) pass 1/2, no budget (build/sgpr100) pass 1/2, mean change'%bits)
    for c in t[('new',1)]:
        a=[t[('new',p)][c] for p in (1,2)]; b=[t[('old',p)][c] for p in (1,2)]
        print('  %-8s %.4f %.4f | %.4f %.4f | %+.1f %%'%(c,a[0],a[1],b[0],b[1],100*(sum(a)/sum(b)-1)))
PY
