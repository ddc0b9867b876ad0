#!/bin/bash
# is the container's CPU quota (cgroup cpu.max) what makes some gapped calls 20-40 ms longer?  cpu.stat before / after, with and without a compact affinity mask
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r6_thr
stat() { grep -E "nr_throttled|throttled_usec|nr_periods" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo; }
run() { # name, prefix...
  local name=$1; shift
  echo "== $name"; stat
  "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cli --no-north-star --no-content --no-pmc > gpurun_out/r6_thr/$name.json 2> gpurun_out/r6_thr/$name.err
  stat
  python -c "
import json; d=json.load(open('gpurun_out/r6_thr/$name.json')); print('ms/step', round(d['ms_per_step'],1), 'gapped alone', [round(x*1e3,1) for x in d['gapped']['wall_s_calls']], 'beside chain', [round(x*1e3,1) for x in d['chain']['wall_s_calls_beside_a_gapped_batch']], 'sbs', round(d['gapped']['wall_s_strand_by_strand']*1e3,1))"
}




run wake1 env LZ_BENCH_WAKE=1
run cold1 env LZ_BENCH_WAKE=0
run wake2 env LZ_BENCH_WAKE=1
run cold2 env LZ_BENCH_WAKE=0
run wake3 env LZ_BENCH_WAKE=1
