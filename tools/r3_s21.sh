#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s21
for i in 1 2; do
  timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli > gpurun_out/s21/b$i.json 2> gpurun_out/s21/b$i.err
  R=$i python - <<'PY'
import json, os
r = os.environ["R"]
d = json.loads(open(f"gpurun_out/s21/b{r}.json").read().strip().splitlines()[-1]); g = d["gapped"]
print("run", r, "wall", round(g["wall_s"], 4), "sbs", round(g["wall_s_strand_by_strand"], 4), "GCUPS", round(g["gcups_wall"], 1), "k_ydrop ms", round(g["k_ydrop_ms"], 1), "cycles/row", round(g["longest_dp"]["cycles_per_row"]), "ok", g.get("alignments_ok"))
PY
done
