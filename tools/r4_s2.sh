#!/bin/bash
# round 4, GPU session 2: gapped stage after the closed-form gap scan + batched coverage checks
set -u
O=gpurun_out/r4_s2; mkdir -p $O
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
export LZGPU_REQUIRE_GPU=1
timeout 900 python -m pytest tests/test_gpu_gapped.py tests/test_gpu_lastz_cli.py -m gpu -x -q > $O/pytest_gapped.txt 2>&1; tail -5 $O/pytest_gapped.txt
for k in 1 2; do
LZGPU_HOSTPROF=1 timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cli > $O/bench_$k.json 2> $O/bench_$k.err
grep -a "hostprof\] gapped" $O/bench_$k.err | tail -2
python - <<PY
import json
d = json.load(open("$O/bench_$k.json")); g = d["gapped"]
print("ms/step", round(d["ms_per_step"],1), "| gapped wall", round(g["wall_s"]*1e3,1), "ms  strand-by-strand", round(g["wall_s_strand_by_strand"]*1e3,1), "GCUPS", round(g["gcups_wall"],1), "k_ydrop", round(g["k_ydrop_ms"],1), g["k_ydrop_launches"], "cyc/row", round(g["longest_dp"]["cycles_per_row"]), "ok", g.get("alignments_ok"))
PY
done
