#!/bin/bash
# round 6, GPU session 2: does rocprofv3's PC sampling work on this box?  (where k_scan_hits' waves spend their time, instruction by instruction)
set -u
O=gpurun_out/r6_s2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1 LZGPU_SERIAL=1
LEGS="--no-cpu-baseline --no-cli --no-north-star --no-content --no-gapped --no-pmc"
for M in "stochastic cycles 1048576" "host_trap time 100"; do
  set -- $M
  timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $1 --pc-sampling-unit $2 --pc-sampling-interval $3 --kernel-trace --output-format csv -d $O/pcs_$1 -- python bench.py --steps 1 --warmup 0 $LEGS --tlen 20000000 --qlen 20000000 > $O/pcs_$1.out 2> $O/pcs_$1.err
  echo "== $1 rc=$?"; tail -5 $O/pcs_$1.err; find $O/pcs_$1 -type f | head; 
  for f in $(find $O/pcs_$1 -name "*pc_sampling*.csv"); do wc -l $f; head -3 $f; done
done
du -sh $O
