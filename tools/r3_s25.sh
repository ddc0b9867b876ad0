#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s25
timeout 900 python bench.py --north-star --steps 1 --warmup 1 --no-cpu-baseline --no-cli > gpurun_out/s25/ns.json 2> gpurun_out/s25/ns.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/s25/ns.json").read().strip().splitlines()[-1]); g = d["gapped"]
print("north star: ms/step", round(d["ms_per_step"], 1), "value", round(d["value"], 4), "frac", round(d["roofline"]["frac"], 3), {k: round(v) for k, v in d["kernel_ms_per_step"].items() if v > 50})
print("gapped wall", round(g["wall_s"], 3), "sbs", round(g["wall_s_strand_by_strand"], 3), "GCUPS", round(g["gcups_wall"], 1), "k_ydrop ms", round(g["k_ydrop_ms"], 1), "launches", g["k_ydrop_launches"], "ok", g.get("alignments_ok"), g["alignments"])
PY
