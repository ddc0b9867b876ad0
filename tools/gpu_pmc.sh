#!/bin/bash
# PMC passes (counters only, --kernel-trace, no other trace domains) of a small bench run; per-kernel sums.
# usage: bash tools/gpu_pmc.sh <tag> [tlen] [qlen]
set -u
TAG=${1:-pmc}; TL=${2:-20000000}; QL=${3:-20000000}
O=gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_SERIAL=1
run() { timeout 300 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $O/p_$2 -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --tlen $TL --qlen $QL > /dev/null 2> $O/err_$2.txt; python tools/pmc_agg.py "$O/p_$2/**/*counter_collection.csv" | grep -v "rocprim\|__amd" > $O/pmc_$2.txt; rm -rf $O/p_$2; }
run "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" a
run "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU" b
run "GRBM_GUI_ACTIVE SQ_IFETCH SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" c
cat $O/pmc_a.txt $O/pmc_b.txt $O/pmc_c.txt
