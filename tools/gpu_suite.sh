#!/bin/bash
# the whole GPU test suite (+ optionally a bench line): bash tools/gpu_suite.sh <outdir> [bench args...]
set -u
O=$1; shift; mkdir -p $O
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
export LZGPU_REQUIRE_GPU=1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt
if [ $# -gt 0 ]; then
  timeout 900 python bench.py "$@" > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.err
  python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("ms/step", round(d["ms_per_step"],1), "frac", round(d["roofline"]["frac"],3), {k: v for k, v in d["parity"].items() if k.endswith("_ok")})
if "gapped" in d: print("gapped", {k: d["gapped"][k] for k in ("wall_s","gcups_wall","k_ydrop_ms","alignments_ok")}, d["gapped"]["longest_dp"]["cycles_per_row"])
if "chain" in d: print("chain", d["chain"])
if "cli" in d: print("cli", d["cli"]["runs_s"])
if "content" in d: print("content", {k: (round(v["ms_per_step"],1), v["scan_mode"], round(v["k_scan_hits"]["frac"],3)) for k, v in d["content"].items()})
ns = d.get("north_star")
if ns: print("north_star", round(ns["ms_per_step"],1), "frac", round(ns["roofline"]["frac"],3), ns["parity"].get("hsp_sha_ok"), round(ns["gapped"]["gcups_wall"],1))
PY
fi
