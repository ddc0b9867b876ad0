#!/bin/bash
set -u
O=gpurun_out/${1:-s4}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
L=$GRAFT_REPO_ROOT/lastz_amd
bash tools/ab.sh "A=base" "LZGPU_LIB=$L/liblzgpu_xnl.so" "LZGPU_LIB=$L/liblzgpu_xnt.so" "LZGPU_LIB=$L/liblzgpu_xloc.so" 2>&1 | tee $O/ab.txt
