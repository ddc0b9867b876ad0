#!/bin/bash
# session 11: format merge tests on two ranks, base tests, CLI timeline with the host phases
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s11
timeout 900 python -m pytest tests/test_gpu_multi.py -x -q -m gpu -k "line_oriented or two_ranks" > gpurun_out/s11/multi.txt 2>&1; tail -5 gpurun_out/s11/multi.txt
timeout 600 python -m pytest tests/test_gpu_seed.py tests/test_gpu_lastz_cli.py -x -q -m gpu > gpurun_out/s11/seed.txt 2>&1; tail -3 gpurun_out/s11/seed.txt
bash tools/cli_prof.sh > gpurun_out/s11/cli_prof.txt 2>&1; tail -60 gpurun_out/s11/cli_prof.txt
