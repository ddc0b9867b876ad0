#!/bin/bash
# host side of the gapped stage after the O(log n) searches: bench pair and north star
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s28
LZGPU_HOSTPROF=1 timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli > gpurun_out/s28/b.json 2> gpurun_out/s28/b.err
grep "gapped:" gpurun_out/s28/b.err | head -4 | cut -c1-230
python - <<'PY'
import json
d = json.loads(open("gpurun_out/s28/b.json").read().strip().splitlines()[-1]); g = d["gapped"]
print("bench pair: wall", round(g["wall_s"], 4), "sbs", round(g["wall_s_strand_by_strand"], 4), "GCUPS", round(g["gcups_wall"], 1), "k_ydrop ms", round(g["k_ydrop_ms"], 1), "ok", g.get("alignments_ok"))
PY
LZGPU_HOSTPROF=1 timeout 900 python bench.py --north-star --steps 1 --warmup 0 --no-cpu-baseline --no-cli > gpurun_out/s28/ns.json 2> gpurun_out/s28/ns.err
grep "gapped:" gpurun_out/s28/ns.err | head -4 | cut -c1-230
python - <<'PY'
import json
d = json.loads(open("gpurun_out/s28/ns.json").read().strip().splitlines()[-1]); g = d["gapped"]
print("north star: wall", round(g["wall_s"], 4), "sbs", round(g["wall_s_strand_by_strand"], 4), "GCUPS", round(g["gcups_wall"], 1), "k_ydrop ms", round(g["k_ydrop_ms"], 1), "alignments", g["alignments"])
PY
