#!/bin/bash
# round 5, GPU session 5: k_scan_tasks with 512 / 1024 lanes per workgroup; the bounded DP kernel without stamp reads on mask-free rows
set -u
O=gpurun_out/r5_s5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
STEPS=4 bash tools/ab_lib.sh $O default r5_st512 r5_st1024 default 2>&1 | tee $O/ab_seed.txt
LZGPU_DPPROF=1 STEPS=2 BENCH_ARGS=" " bash tools/ab_lib.sh $O default 2>&1 | tee $O/ab_dp.txt
grep "dpprof\] launch" $O/bench_default.err | tail -4
timeout 600 python -m pytest tests/test_gpu_gapped.py tests/test_gpu_seed.py -m gpu -x -q -k "not north_star and not full_size" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
