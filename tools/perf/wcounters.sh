#!/bin/bash
# TCP -> L2 request counters of write patterns (one pattern per process, one counter group per pass).
# In flight per CU = LATENCY_sum / (GRBM_GUI_ACTIVE / 8 XCDs) / 256 CUs.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/wcounters; rm -rf $out; mkdir -p $out
patterns=("steps 1 256" "steps 3 256" "tiles 8" "tiles 16" "tiles 32" "gather 32 0" "gather 32 1" "gather 8 1")
sets=("TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum"
      "GRBM_GUI_ACTIVE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"
      "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_sum TCC_TAG_STALL_sum")
i=0
for set in "${sets[@]}"; do
  j=0
  for pattern in "${patterns[@]}"; do
    rocprofv3 --pmc $set --output-format csv -d $out/s${i}_p$j -o pmc -- tools/perf/wsingle $pattern > $out/s${i}_p$j.log 2>&1
    j=$((j+1))
  done
  i=$((i+1))
done
python3 - <<PY
import csv, collections, glob
patterns="steps 1 256|steps 3 256|tiles 8|tiles 16|tiles 32|gather 32 0|gather 32 1|gather 8 1".split('|')
table=collections.defaultdict(dict)
for f in sorted(glob.glob('$out/*/*counter_collection.csv')):
    p=int(f.split('/')[2].split('_p')[1])
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): table[p][k]=sum(v)/len(v)
for p in sorted(table):
    t=table[p]; cyc=t.get('GRBM_GUI_ACTIVE',0)/8
    ms=open('$out/s0_p%d.log'%p).read().strip().splitlines()[-1]
    print('%-14s %s' % (patterns[p], ms))
    if cyc:
        wl=t['TCP_TCC_WRITE_REQ_LATENCY_sum']; wn=t['TCP_TCC_WRITE_REQ_sum']; rl=t['TCP_TCC_READ_REQ_LATENCY_sum']; rn=t['TCP_TCC_READ_REQ_sum']
        print('   cycles/XCD %.3g | writes %.4g avg latency %.0f in flight/CU %.1f | reads %.4g avg latency %.0f in flight/CU %.1f' % (cyc, wn, wl/max(wn,1), wl/cyc/256, rn, rl/max(rn,1), rl/cyc/256))
        print('   L2->fabric: write reqs %.4g in flight/channel %.1f dram-credit stall %.1f%% | read reqs %.4g in flight/channel %.1f | TCC busy %.0f%% tag stall %.1f%%' % (
            t['TCC_EA0_WRREQ_sum'], t['TCC_EA0_WRREQ_LEVEL_sum']/cyc/128, 100*t['TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum']/cyc/128, t['TCC_EA0_RDREQ_sum'], t['TCC_EA0_RDREQ_LEVEL_sum']/cyc/128, 100*t['TCC_BUSY_sum']/cyc/128, 100*t['TCC_TAG_STALL_sum']/cyc/128))
PY
