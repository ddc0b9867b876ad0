#!/bin/bash
# round 5, GPU session 9: the serial piece of the bounded DP replicated on all four waves (small launches; forced on / off), per-launch clocks
set -u
O=gpurun_out/r5_s9; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
timeout 900 python -m pytest tests/test_gpu_gapped.py tests/test_gpu_base_tests.py -m gpu -x -q > $O/pytest_a.txt 2>&1; tail -2 $O/pytest_a.txt
LZGPU_DP_REPL=1 LZGPU_DP_HORIZON=97 timeout 900 python -m pytest tests/test_gpu_gapped.py tests/test_gpu_lastz_cli.py -m gpu -x -q > $O/pytest_repl1.txt 2>&1; tail -2 $O/pytest_repl1.txt
LZGPU_DP_REPL=0 timeout 900 python -m pytest tests/test_gpu_gapped.py -m gpu -x -q > $O/pytest_repl0.txt 2>&1; tail -2 $O/pytest_repl0.txt
LZGPU_DPPROF=1 STEPS=2 BENCH_ARGS=" " bash tools/ab_lib.sh $O default 2>&1 | tee $O/ab.txt
grep "dpprof\] launch" $O/bench_default.err | tail -4
