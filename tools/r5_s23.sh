#!/bin/bash
# round 5, GPU session 23: the driver's bench command with the gapped leg's wall as the median of three calls
set -u
O=gpurun_out/r05_f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; echo
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r05_f/bench.json").read().split("\n") if l.startswith("{")][-1])
g=d["gapped"]; print("ms/step", d["ms_per_step"], "gapped", g["wall_s"], g["wall_s_calls"], g["gcups_wall"], "NS", d["north_star"]["ms_per_step"], d["north_star"]["gapped"]["wall_s"], d["north_star"]["gapped"]["wall_s_calls"], d["north_star"]["gapped"]["gcups_wall"])
PY
timeout 600 python -m pytest tests/test_gpu_multi.py -m gpu -x -q -k "bench" 2>&1 | tail -2
