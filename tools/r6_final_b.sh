#!/bin/bash
# round 6, last pass on the final tree: the driver's bench command first (a fresh box's first process), then the GPU suite and smoke
set -u
O=gpurun_out/r06_b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 1600 $O/bench.json; echo; wc -l $O/bench.json
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
