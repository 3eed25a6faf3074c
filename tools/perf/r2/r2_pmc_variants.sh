echo "== default (R=8, 32 copies)"; bash tools/perf/pmc_split.sh r2_6bit_c32 fasttext2m-300d-6bit-fullvocab 2>&1 | grep flags
echo "== R=10"; MEMB_HIP_BYTE_ROOT_BITS=10 bash tools/perf/pmc_split.sh r2_6bit_r10 fasttext2m-300d-6bit-fullvocab 2>&1 | grep flags
echo "== R=10, 1 copy"; MEMB_HIP_BYTE_ROOT_BITS=10 MEMB_HIP_TABLE_COPIES=1 bash tools/perf/pmc_split.sh r2_6bit_r10c1 fasttext2m-300d-6bit-fullvocab 2>&1 | grep flags
