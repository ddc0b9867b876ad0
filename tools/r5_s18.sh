#!/bin/bash
# round 5, GPU session 18: does where the big buffers lie decide the fill / partition kernels' process-to-process spread?  (addresses printed by
# LZGPU_HOSTPROF; LZGPU_ALLOC_ALIGN=<log2> rounds them up to multiples of 2^n bytes)
set -u
O=gpurun_out/r5_s18; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for k in 1 2 3; do
  LZGPU_HOSTPROF=1 STEPS=3 bash tools/ab_lib.sh $O/$k default default:LZGPU_ALLOC_ALIGN=21 default:LZGPU_ALLOC_ALIGN=30 2>&1 | cut -c1-300
done | tee $O/ab.txt
for f in $O/1/bench_default.err $O/2/bench_default.err $O/1/bench_default_LZGPU_ALLOC_ALIGN_21.err; do echo $f; grep "device buffer" $f | awk '{print $5,$6,$7,$8,$9}' | sort | uniq -c | sort -rn | head -30; done > $O/addresses.txt
