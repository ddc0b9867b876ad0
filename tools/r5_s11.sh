#!/bin/bash
# round 5, GPU session 11: the partition bytes written by the scan kernel (coalesced) instead of the fill kernel (39-byte runs): seed tests, then
# the bench line five times
set -u
O=gpurun_out/r5_s11; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
timeout 900 python -m pytest tests/test_gpu_seed.py -m gpu -x -q -k "not north_star" > $O/pytest_seed.txt 2>&1; tail -2 $O/pytest_seed.txt
STEPS=3 bash tools/ab_lib.sh $O default default default default default 2>&1 | cut -c1-250 | tee $O/ab.txt
bash tools/ab_ns.sh $O default 2>&1 | tee $O/ab_ns.txt
