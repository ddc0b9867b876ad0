#!/bin/bash
# gap rule of the speculation windows: bench gapped leg with and without, tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s17
for rule in 1 0 1 0; do
  if [ $rule = 0 ]; then export LZGPU_DP_NO_GAP_RULE=1; else unset LZGPU_DP_NO_GAP_RULE; fi
  timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli > gpurun_out/s17/b$rule.json 2> gpurun_out/s17/b$rule.err
  R=$rule python - <<'PY'
import json, os
r = os.environ["R"]
d = json.loads(open(f"gpurun_out/s17/b{r}.json").read().strip().splitlines()[-1]); g = d["gapped"]
print("gap rule", r, "wall", round(g["wall_s"], 4), "strand by strand", round(g["wall_s_strand_by_strand"], 4), "GCUPS", round(g["gcups_wall"], 1), "k_ydrop ms", round(g["k_ydrop_ms"], 1), "launches", g["k_ydrop_launches"], "launched", g["dp_launched"], "extended", g["anchors_extended"], "ok", g.get("alignments_ok"))
PY
done
unset LZGPU_DP_NO_GAP_RULE
LZGPU_HOSTPROF=1 timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli 2>&1 >/dev/null | grep "window cut\|speculates\|gapped:" | head -8 | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_gapped.py tests/test_gpu_lastz_cli.py -x -q -m gpu 2>&1 | tail -3
