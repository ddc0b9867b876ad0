#!/bin/bash
# round 5, GPU session 16: do the two kernels of a mixed DP launch overlap?  (HIP maps streams onto GPU_MAX_HW_QUEUES = 4 hardware queues)
set -u
O=gpurun_out/r5_s16; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
LZGPU_DPPROF=1 STEPS=2 BENCH_ARGS=" " bash tools/ab_lib.sh $O default:GPU_MAX_HW_QUEUES=8 default:GPU_MAX_HW_QUEUES=8,LZGPU_DP_LONG=1.2 default:GPU_MAX_HW_QUEUES=2 2>&1 | cut -c1-30,250-400 | tee $O/ab.txt
for f in $O/bench_*.err; do echo $f; grep "dpprof\]" $f | head -2 | cut -c1-300; done
