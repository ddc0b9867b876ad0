#!/bin/bash
set -u
O=gpurun_out/r4_s4; mkdir -p $O
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
LZGPU_LIB=$PWD/lastz_amd/liblzgpu_clk.so timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cli --no-gapped > $O/bench_clk.json 2> $O/bench_clk.err
grep -a "phase clocks" $O/bench_clk.err
python -c "
import json; d=json.load(open('$O/bench_clk.json')); print(round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['kernel_ms_per_step'].items() if v>1.5}, d['counters_per_step'])"
