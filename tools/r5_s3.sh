#!/bin/bash
# round 5, GPU session 3: the DP with bounds / masks from host-made pieces -- gapped tests, the reference's base tests and the CLI cases through
# the bound binary, the same with a 97-row first horizon (re-runs), the bench's gapped leg with per-launch clocks
set -u
O=gpurun_out/r5_s3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
timeout 1200 python -m pytest tests/test_gpu_gapped.py tests/test_gpu_base_tests.py tests/test_gpu_lastz_cli.py -m gpu -x -q > $O/pytest_a.txt 2>&1; tail -4 $O/pytest_a.txt
LZGPU_DP_HORIZON=97 timeout 900 python -m pytest tests/test_gpu_gapped.py tests/test_gpu_base_tests.py -m gpu -x -q > $O/pytest_h97.txt 2>&1; tail -4 $O/pytest_h97.txt
LZGPU_DPPROF=1 STEPS=2 BENCH_ARGS=" " bash tools/ab_lib.sh $O default 2>&1 | tee $O/ab.txt
grep "dpprof\] launch" $O/bench_default.err | tail -8
