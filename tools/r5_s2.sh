#!/bin/bash
# round 5, GPU session 2: FETCH_SIZE calibration, the N > 1 path's tests, the north-star test with the LAV fingerprint, the full default bench line
set -u
O=gpurun_out/r5_s2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
bash tools/fetch_calib.sh $O/calib > $O/calib.txt 2>&1; tail -9 $O/calib.txt
cp $O/calib/fetch_size_calibration.json profiles/fetch_size_calibration.json 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_multi.py -m gpu -x -q -k "bench" > $O/pytest_multi.txt 2>&1; tail -5 $O/pytest_multi.txt
timeout 600 python -m pytest tests/test_gpu_seed.py -m gpu -x -q -k "north_star" > $O/pytest_ns.txt 2>&1; tail -3 $O/pytest_ns.txt
timeout 1500 python bench.py --gpus 1 --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json; echo; tail -5 $O/bench.err
