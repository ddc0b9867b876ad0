#!/bin/bash
# SQ counters of the seed kernels for one library build: bash tools/r6_pmc.sh <outdir> <tag|default> [size]
set -u
O=$1; TAG=$2; S=${3:-50000000}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
lib=$PWD/lastz_amd/liblzgpu.so; [ "$TAG" != "default" ] && lib=$PWD/lastz_amd/liblzgpu_$TAG.so
export LZGPU_LIB=$lib LZGPU_SERIAL=1
LEGS="--no-cpu-baseline --no-cli --no-north-star --no-content --no-gapped --no-pmc"
pmc() { local tag=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/p_$tag -- python bench.py --steps 1 --warmup 0 $LEGS --tlen $S --qlen $S > /dev/null 2> $O/err_$tag.txt
  python tools/pmc_agg.py "$O/p_$tag/**/*counter_collection.csv" | grep "${KERN:-k_scan_hits}" > $O/pmc_${TAG}_$tag.txt; rm -rf $O/p_$tag; cat $O/pmc_${TAG}_$tag.txt | cut -c1-500; }
pmc a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pmc b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU
pmc c GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_BRANCH
