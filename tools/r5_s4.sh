#!/bin/bash
# round 5, GPU session 4: DP pieces after the horizon fix (default, 97- and 776-row first horizons), per-launch clocks of the bounded kernel,
# host profile of the gapped batch
set -u
O=gpurun_out/r5_s4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
timeout 900 python -m pytest tests/test_gpu_gapped.py -m gpu -x -q > $O/pytest_a.txt 2>&1; tail -2 $O/pytest_a.txt
LZGPU_DP_HORIZON=97 timeout 900 python -m pytest tests/test_gpu_gapped.py tests/test_gpu_base_tests.py -m gpu -x -q > $O/pytest_h97.txt 2>&1; tail -2 $O/pytest_h97.txt
LZGPU_DP_HORIZON=776 timeout 900 python -m pytest tests/test_gpu_gapped.py tests/test_gpu_lastz_cli.py -m gpu -x -q > $O/pytest_h776.txt 2>&1; tail -2 $O/pytest_h776.txt
LZGPU_DPPROF=1 LZGPU_HOSTPROF=1 STEPS=2 BENCH_ARGS=" " bash tools/ab_lib.sh $O default 2>&1 | tee $O/ab.txt
grep "dpprof\] launch\|hostprof\] gapped" $O/bench_default.err | tail -12
