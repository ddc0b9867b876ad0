#!/bin/bash
set -u
O=gpurun_out/${1:-s10}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
L=$GRAFT_REPO_ROOT/lastz_amd
bash tools/ab.sh "A=base" "LZGPU_HIT_CAPACITY=2147483648" "LZGPU_LIB=$L/liblzgpu_pp1024.so" "LZGPU_LIB=$L/liblzgpu_pp1024t16.so" 2>&1 | tee $O/ab.txt
