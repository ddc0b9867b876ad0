#!/bin/bash
set -u
O=gpurun_out/${1:-s7}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
timeout 1500 python -m pytest tests/test_gpu_gapped.py -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest.txt
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cli > $O/bench.json 2> $O/bench.err; python - <<PY
import json
d=json.load(open("$O/bench.json")); g=d["gapped"]
print("ms/step", d["ms_per_step"], "gapped wall", g["wall_s"], "seq", g["wall_s_strand_by_strand"], "gcups", g["gcups_wall"], "k_ydrop ms", g["k_ydrop_ms"], g["k_ydrop_launches"], g["alignments_ok"], g["longest_dp"])
PY
tail -3 $O/bench.err
LZGPU_DP_UNIFORM_SLOTS=1 timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli > $O/bench_u.json 2> $O/bench_u.err; python - <<PY
import json
d=json.load(open("$O/bench_u.json")); g=d["gapped"]
print("UNIFORM: gapped wall", g["wall_s"], "seq", g["wall_s_strand_by_strand"], "k_ydrop ms", g["k_ydrop_ms"], g["k_ydrop_launches"], g["alignments_ok"])
PY
