#!/bin/bash
# round 6, GPU session 1: (a) bench.py's N > 1 branch over RCCL with one rank (the new -m gpu test), (b) a short bench line as the
# baseline of the round, (c) SQ / TCP / TCC counters of the CURRENT k_scan_hits at 50 and 200 Mbp (VERDICT r5 #1a: measure first)
set -u
O=gpurun_out/r6_s1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
timeout 900 python -m pytest tests/test_gpu_multi.py -m gpu -x -q -k "rccl or two_ranks_search" > $O/pytest_rccl.txt 2>&1; tail -5 $O/pytest_rccl.txt
LEGS="--no-cpu-baseline --no-cli --no-north-star --no-content"
timeout 600 python bench.py --steps 10 --warmup 3 $LEGS > $O/bench_50m.json 2> $O/bench_50m.err; tail -c 1200 $O/bench_50m.json; echo
pmc() { # tag size counters...
  local tag=$1 size=$2; shift 2
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/p_$tag -- python bench.py --steps 1 --warmup 0 $LEGS --no-gapped --no-pmc --tlen $size --qlen $size > /dev/null 2> $O/err_$tag.txt
  python tools/pmc_agg.py "$O/p_$tag/**/*counter_collection.csv" | grep "k_scan_hits\|k_partition\|k_settle\|k_fill" > $O/pmc_$tag.txt; rm -rf $O/p_$tag; cat $O/pmc_$tag.txt | cut -c1-400
}
for S in 50000000 200000000; do
  T=$((S/1000000))m
  LZGPU_SERIAL=1 pmc a_$T $S SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  LZGPU_SERIAL=1 pmc b_$T $S SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU
  LZGPU_SERIAL=1 pmc c_$T $S GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_BRANCH
  LZGPU_SERIAL=1 pmc d_$T $S TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
  LZGPU_SERIAL=1 pmc e_$T $S TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum
done
