#!/bin/bash
# round 5, GPU session 1: the ADVICE fixes under the gapped tests, then A/B lines: DP with 128 lanes (and a 1024-column ring),
# seed-stage what-ifs (no key loads in the scan kernel, no key stores in the fill kernel, L2-resident target at 200 Mbp)
set -u
O=gpurun_out/r5_s1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
timeout 400 python -m pytest tests/test_gpu_gapped.py -m gpu -x -q > $O/pytest_gapped.txt 2>&1; tail -3 $O/pytest_gapped.txt
STEPS=3 BENCH_ARGS=" " bash tools/ab_lib.sh $O default r5_dp128 r5_dp128r1k 2>&1 | tee $O/ab_dp.txt
STEPS=3 bash tools/ab_lib.sh $O r5_nokeyld r5_nokeyst 2>&1 | tee $O/ab_seed.txt
bash tools/ab_ns.sh $O default r5_localt 2>&1 | tee $O/ab_ns.txt
