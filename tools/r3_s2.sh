#!/bin/bash
# round 3, GPU session 2: occupancy variants of k_scan_hits<0>, geometry variants of k_settle2
set -u
O=gpurun_out/${1:-s2}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
L=$GRAFT_REPO_ROOT/lastz_amd
bash tools/ab.sh "LZGPU_SC_TPB=512" "LZGPU_SC_TPB=640" "LZGPU_SC_TPB=768" "LZGPU_SC_TPB=1024" \
  "LZGPU_LIB=$L/liblzgpu_w8.so" "LZGPU_LIB=$L/liblzgpu_r6.so" "LZGPU_LIB=$L/liblzgpu_w8r6.so" "LZGPU_LIB=$L/liblzgpu_r3.so" 2>&1 | tee $O/ab.txt
