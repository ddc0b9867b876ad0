#!/bin/bash
# round 5, GPU session 7: the sorted list word-sorted inside blocks of positions (LZGPU_BLOCK_BITS=4, default) against one word order (0):
# seed tests, then the same binary alternately, then the 200 Mbp pair
set -u
O=gpurun_out/r5_s7; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
timeout 900 python -m pytest tests/test_gpu_seed.py -m gpu -x -q -k "not north_star" > $O/pytest_seed.txt 2>&1; tail -2 $O/pytest_seed.txt
STEPS=3 bash tools/ab_lib.sh $O default:LZGPU_BLOCK_BITS=0 default default:LZGPU_BLOCK_BITS=0 default default:LZGPU_BLOCK_BITS=2 default:LZGPU_BLOCK_BITS=6 default 2>&1 | tee $O/ab.txt
bash tools/ab_ns.sh $O default 2>&1 | tee $O/ab_ns.txt
