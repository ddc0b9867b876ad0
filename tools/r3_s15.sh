#!/bin/bash
# north-star gapped leg against the speculation window (anchors per round)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s15
for w in 2048 8192 32768; do
  LZGPU_DP_WINDOW=$w timeout 900 python bench.py --north-star --steps 1 --warmup 0 --no-cpu-baseline --no-cli > gpurun_out/s15/w$w.json 2> gpurun_out/s15/w$w.err
  W=$w python - <<'PY'
import json, os
w = os.environ["W"]
try:
    d = json.loads(open(f"gpurun_out/s15/w{w}.json").read().strip().splitlines()[-1]); g = d["gapped"]
    print("window", w, "wall", round(g["wall_s"], 3), "strand by strand", round(g["wall_s_strand_by_strand"], 3), "GCUPS", round(g["gcups_wall"], 1), "k_ydrop ms", round(g["k_ydrop_ms"], 1), "launches", g["k_ydrop_launches"], "launched", g["dp_launched"], "ok", g["alignments_ok"])
except Exception as e:
    print("window", w, "failed", e, open(f"gpurun_out/s15/w{w}.err").read()[-500:])
PY
done
timeout 900 python -m pytest tests/test_gpu_gapped.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_lastz_cli.py tests/test_gpu_base_tests.py -x -q -m gpu 2>&1 | tail -3
