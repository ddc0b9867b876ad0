#!/bin/bash
set -u
O=gpurun_out/r4_s6; mkdir -p $O
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
export LZGPU_REQUIRE_GPU=1
timeout 1200 python -m pytest tests/test_gpu_multi.py -m gpu -x -q > $O/pytest_multi.txt 2>&1; tail -15 $O/pytest_multi.txt
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_s6/bench_default.json"))
print("ms/step", round(d["ms_per_step"],1), "frac", round(d["roofline"]["frac"],3), d["parity"])
print("gapped", {k: d["gapped"][k] for k in ("wall_s","gcups_wall","k_ydrop_ms","alignments_ok")}, d["gapped"]["longest_dp"])
print("cli", d.get("cli"))
ns = d.get("north_star")
if ns: print("north_star", round(ns["ms_per_step"],1), "frac", round(ns["roofline"]["frac"],3), ns["parity"], {k: ns["gapped"][k] for k in ("wall_s","gcups_wall","k_ydrop_ms","alignments")}, ns.get("cli"))
print("content", json.dumps(d.get("content")))
PY
