#!/bin/bash
# round 5, GPU session 25: the gapped / CLI / base-test files with the two-wave DP kernel forced for every launch, one leading wave per DP and a copy of the
# row set-up in both waves (the launcher's rule picks the two-wave kernel only for launches of thousands of DPs, and its one-leading-wave form beyond 5632)
set -u
O=gpurun_out/r5_s25; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
for r in 0 1; do
  LZGPU_DP_NARROW=1 LZGPU_DP_REPL=$r timeout 900 python -m pytest tests/test_gpu_lastz_cli.py tests/test_gpu_base_tests.py tests/test_gpu_gapped.py -m gpu -x -q > $O/pytest_narrow1_repl$r.txt 2>&1; echo "NARROW=1 REPL=$r: $(tail -1 $O/pytest_narrow1_repl$r.txt)"
done | tee $O/summary.txt
