#!/bin/bash
# round 5, GPU session 14: the two-wave / one-wave DP kernel with the 16-bit sweep row (dp_kernels_narrow.hip) against the four-wave kernel;
# with and without the priority steps of long sweeps
set -u
O=gpurun_out/r5_s14; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
LZGPU_DP_NARROW=1 timeout 600 python -m pytest tests/test_gpu_gapped.py -m gpu -x -q > $O/pytest_narrow1.txt 2>&1; tail -3 $O/pytest_narrow1.txt
LZGPU_DPPROF=1 STEPS=2 BENCH_ARGS=" " bash tools/ab_lib.sh $O default:LZGPU_DP_NARROW=0 default r5_noprio r5_n64 r5_n64noprio r5_n64b8 2>&1 | cut -c1-400 | tee $O/ab.txt
for f in $O/bench_*.err; do echo $f; grep "dpprof\] launch" $f | head -3 | cut -c1-300; done
