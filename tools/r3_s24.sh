#!/bin/bash
cd $GRAFT_REPO_ROOT
LZGPU_DPPROF=1 timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli 2>&1 >/dev/null | grep "dpprof\] launch" | cut -c1-200
