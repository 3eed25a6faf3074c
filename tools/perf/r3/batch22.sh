#!/bin/bash
# round 3, batch 22: L2 -> fabric queue counters of the two large-batch kernels on one box (4-bit and 2-bit dumps)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r3/b22; mkdir -p $out
export MEMB_HIP_AUTOTUNE=0
for workload in glove840b-300d-4bit-fullvocab glove840b-300d-2bit-fullvocab; do
for kind in 2 0; do
  export MEMB_HIP_PERSISTENT=$kind
  args="bench.py --workload $workload --no-configs --no-cpu-baseline --steps 5 --warmup 2"
  rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum --output-format csv -d $out/${workload}_p${kind}_a -o pmc -- python3 $args > /dev/null 2> $out/${workload}_p${kind}_a.err
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_sum TCC_TAG_STALL_sum --output-format csv -d $out/${workload}_p${kind}_b -o pmc -- python3 $args > /dev/null 2> $out/${workload}_p${kind}_b.err
  rocprofv3 --pmc TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $out/${workload}_p${kind}_c -o pmc -- python3 $args > /dev/null 2> $out/${workload}_p${kind}_c.err
done
done
python3 - <<'PY'
import csv, collections, glob
out='gpurun_out/r3/b22'
for workload in ('glove840b-300d-4bit-fullvocab','glove840b-300d-2bit-fullvocab'):
    for kind,name in ((2,'persistent'),(0,'one tile per wavefront')):
        t={}
        for f in glob.glob('%s/%s_p%d_*/*counter_collection.csv'%(out,workload,kind)):
            agg=collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if 'decode_trained' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
            for k,v in agg.items(): t[k]=sum(v)/len(v)
        cyc=t['GRBM_GUI_ACTIVE']/8
        print('%s | %s' % (workload, name))
        print('   cycles/XCD %.4g | TCP->L2 writes %.4g avg latency %.0f in flight/CU %.1f | reads %.4g avg latency %.0f in flight/CU %.1f' % (cyc, t['TCP_TCC_WRITE_REQ_sum'], t['TCP_TCC_WRITE_REQ_LATENCY_sum']/t['TCP_TCC_WRITE_REQ_sum'], t['TCP_TCC_WRITE_REQ_LATENCY_sum']/cyc/256, t['TCP_TCC_READ_REQ_sum'], t['TCP_TCC_READ_REQ_LATENCY_sum']/t['TCP_TCC_READ_REQ_sum'], t['TCP_TCC_READ_REQ_LATENCY_sum']/cyc/256))
        print('   L2->fabric: write reqs %.4g in flight/channel %.1f dram-credit stall %.1f%% | read reqs %.4g in flight/channel %.1f | TCC busy %.0f%% tag stall %.1f%%' % (t['TCC_EA0_WRREQ_sum'], t['TCC_EA0_WRREQ_LEVEL_sum']/cyc/128, 100*t['TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum']/cyc/128, t['TCC_EA0_RDREQ_sum'], t['TCC_EA0_RDREQ_LEVEL_sum']/cyc/128, 100*t['TCC_BUSY_sum']/cyc/128, 100*t['TCC_TAG_STALL_sum']/cyc/128))
PY
