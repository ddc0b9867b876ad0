#!/bin/bash
# why the gapped stage of the bench pair needs more than one round: the anchors the windows are cut at
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s16
LZGPU_HOSTPROF=1 timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli > gpurun_out/s16/b.json 2> gpurun_out/s16/b.err
grep "window cut\|speculates\|gapped:" gpurun_out/s16/b.err | cut -c1-220
