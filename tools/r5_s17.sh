#!/bin/bash
# round 5, GPU session 17: the two-wave kernel with one leading wave per DP against a copy of the row set-up in both waves (the batch of 4596 DPs);
# the whole gapped / CLI / multi test files on the default rule
set -u
O=gpurun_out/r5_s17; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
STEPS=2 BENCH_ARGS=" " bash tools/ab_lib.sh $O default default:LZGPU_DP_REPL=0 default:LZGPU_DP_REPL=1 default default:LZGPU_DP_REPL=0 2>&1 | cut -c1-40,240-400 | tee $O/ab.txt
timeout 900 python -m pytest tests/test_gpu_gapped.py tests/test_gpu_lastz_cli.py tests/test_gpu_base_tests.py -m gpu -x -q > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
