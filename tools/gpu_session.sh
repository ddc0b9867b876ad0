#!/bin/bash
# One GPU-box pass for a development iteration: seed-stage parity tests, a short bench line (overlapped and with
# the streams serialised for clean per-kernel times) and the rocprofv3 kernel stats of the bench command.
# usage (through gpurun): bash tools/gpu_session.sh <tag> [pytest args...]
set -u
TAG=${1:-dev}; shift || true
O=gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
timeout 1200 python -m pytest "${@:-tests/test_gpu_seed.py}" -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -25 $O/pytest_gpu.txt
timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; tail -5 $O/bench.err
LZGPU_SERIAL=1 timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_serial.json 2> $O/bench_serial.err; python - <<PY
import json
try:
    d = json.load(open("$O/bench_serial.json")); print("SERIAL ms/step", d["ms_per_step"], json.dumps(d["kernel_ms_per_step"]))
except Exception as e: print("serial bench failed", e)
PY
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
head -16 $O/kernel_stats.csv
rm -rf $O/stats
