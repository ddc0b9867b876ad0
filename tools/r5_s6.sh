#!/bin/bash
# round 5, GPU session 6: why k_fill_hits2 takes 18 or 28 ms from process to process on one box -- the same binary five times, with the
# keys of a strand in one chunk (2^31 hits: 15.5 GB scatter range) and in chunks of 2^28 (2 GB)
set -u
O=gpurun_out/r5_s6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
STEPS=3 bash tools/ab_lib.sh $O default default:LZGPU_HIT_CAPACITY=268435456 default default:LZGPU_HIT_CAPACITY=268435456 default default:LZGPU_HIT_CAPACITY=1073741824 2>&1 | tee $O/ab.txt
rocm-smi --showmeminfo vram 2>/dev/null | head -8
