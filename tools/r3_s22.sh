#!/bin/bash
# phase clocks of the longest DP (variant build with -DLZ_DP_PHASE_CLOCKS)
cd $GRAFT_REPO_ROOT
LZGPU_LIB=$GRAFT_REPO_ROOT/lastz_amd/liblzgpu_clk.so LZGPU_DPPROF=1 timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli 2>&1 >/dev/null | grep "dpprof" | head -12 | cut -c1-420
