#!/bin/bash
set -u
O=gpurun_out/${1:-s5}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
timeout 1500 python -m pytest tests/test_gpu_multi.py tests/test_gpu_lastz_cli.py -m gpu -x -q -s 2>&1 | tail -12 | tee $O/pytest.txt
LZ_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --tlen-multi 5000000 --q-units 3 --q-unit-len 3000000 --cpu-sample 1000000 > $O/bench_multi2.json 2> $O/bench_multi2.err; tail -c 2500 $O/bench_multi2.json; tail -3 $O/bench_multi2.err
