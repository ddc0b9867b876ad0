#!/bin/bash
# round 5, GPU session 15: the mixed launch -- the DPs expected to sweep more than LONG x the average slot's rows on the four-wave kernel
# (second stream), the rest on the two-wave kernel -- for several LONG factors (0: two-wave kernel alone), against the four-wave kernel alone
set -u
O=gpurun_out/r5_s15; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
timeout 600 python -m pytest tests/test_gpu_gapped.py -m gpu -x -q -k "both_builds or both_forms or golden" > $O/pytest_a.txt 2>&1; tail -3 $O/pytest_a.txt
LZGPU_DPPROF=1 STEPS=2 BENCH_ARGS=" " bash tools/ab_lib.sh $O default:LZGPU_DP_NARROW=0 default:LZGPU_DP_LONG=0 default default:LZGPU_DP_LONG=1.0 default:LZGPU_DP_LONG=1.2 default:LZGPU_DP_LONG=1.7 default:LZGPU_DP_LONG=2.0 default 2>&1 | cut -c1-30,250-400 | tee $O/ab.txt
for f in $O/bench_*.err; do echo $f; grep "dpprof\] launch" $f | head -1 | cut -c1-260; done
