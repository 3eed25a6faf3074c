#!/bin/bash
# usage: tools/perf/prof.sh <tag> <kernel name substring> <bench.py arguments...>   (repo root, GPU box)
# One workload of bench.py under rocprofv3: kernel trace + stats, then the counter groups one pass each
# (never combined with a trace, FETCH_SIZE and WRITE_SIZE in passes of their own: MI355X_MICROARCH.md),
# summarised into gpurun_out/prof_<tag>/summary.json. Example:
#   tools/perf/prof.sh r06_union decode_union_split --workload union-concat-500k
tag=$1; kernel=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag; mkdir -p $out
long="bench.py $* --no-configs --no-cpu-baseline --no-live-traffic --steps 10 --warmup 3"
short="bench.py $* --no-configs --no-cpu-baseline --no-live-traffic --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- python3 $long > $out/bench.json 2> $out/trace.err || exit 1
pass() {   # name, counters...
    name=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o pmc -- python3 $short > /dev/null 2> $out/$name.err || echo "pass $name failed" >&2
}
pass pmc1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS
pass pmc2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR
pass pmc3 GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM
pass pmc4 TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pass tcc2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
python3 tools/perf/prof_summary.py "$out" "$kernel"
cp $out/trace/*kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
