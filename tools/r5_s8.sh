#!/bin/bash
# round 5, GPU session 8: the whole GPU suite on the current tree
set -u
O=gpurun_out/r5_s8; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -15 $O/pytest_gpu.txt
