#!/bin/bash
# gapped leg of the bench with the parallel pre-commit; gapped + CLI tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s20
for i in 1 2; do
  timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli > gpurun_out/s20/b$i.json 2> gpurun_out/s20/b$i.err
  R=$i python - <<'PY'
import json, os
r = os.environ["R"]
d = json.loads(open(f"gpurun_out/s20/b{r}.json").read().strip().splitlines()[-1]); g = d["gapped"]
print("run", r, "wall", round(g["wall_s"], 4), "strand by strand", round(g["wall_s_strand_by_strand"], 4), "GCUPS", round(g["gcups_wall"], 1), "k_ydrop ms", round(g["k_ydrop_ms"], 1), "launches", g["k_ydrop_launches"], "ok", g.get("alignments_ok"), "seed ms", round(d["ms_per_step"], 1))
PY
done
LZGPU_HOSTPROF=1 timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli 2>&1 >/dev/null | grep "gapped:" | head -6 | cut -c1-220
timeout 900 python -m pytest tests/test_gpu_gapped.py tests/test_gpu_lastz_cli.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -3
