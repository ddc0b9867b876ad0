#!/bin/bash
set -u
O=gpurun_out/${1:-s3}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
L=$GRAFT_REPO_ROOT/lastz_amd
export LZGPU_REQUIRE_GPU=1
timeout 900 python -m pytest tests/test_gpu_seed.py -m gpu -x -q 2>&1 | tail -3
bash tools/ab.sh "A=preload" "LZGPU_SC_TPB=640" "LZGPU_LIB=$L/liblzgpu_pp.so" "LZGPU_LIB=$L/liblzgpu_r6.so" 2>&1 | tee $O/ab.txt
rocprofv3 -L > $O/counters.txt 2>&1
