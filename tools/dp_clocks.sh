#!/bin/bash
# where a row's cycles go: the -DLZ_DP_PHASE_CLOCKS build (LZGPU_BUILD_TAG=dpclk) under LZGPU_DPPROF=1, one leading wave and replicated
# usage: bash tools/dp_clocks.sh <outdir>
set -u
O=$1; mkdir -p $O
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for repl in 0 1; do
  LZGPU_LIB=$PWD/lastz_amd/liblzgpu_dpclk.so LZGPU_DPPROF=1 LZGPU_DP_REPL=$repl timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cli --no-north-star --no-content > $O/bench_repl$repl.json 2> $O/bench_repl$repl.err
  echo "== LZGPU_DP_REPL=$repl"; grep -a "dpprof" $O/bench_repl$repl.err | tail -8 | cut -c1-400
done
