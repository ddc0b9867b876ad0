#!/bin/bash
# the gapped leg at 50 and 200 Mbp for environment settings: bash tools/r6_gapped.sh <outdir> "<ENV=VAL ...>" ...
set -u
O=$1; shift; mkdir -p $O
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for envs in "$@"; do
  name=$(echo "$envs" | tr ' =' '__')
  for size in 50 200; do
    args="--no-cli --no-north-star"; [ $size = 200 ] && args="--north-star"
    env $envs timeout 900 python bench.py $args --steps 2 --warmup 1 --no-cpu-baseline --no-content --no-pmc > $O/g_${name}_$size.json 2> $O/g_${name}_$size.err
    python - <<PY
import json
try:
    d = json.load(open("$O/g_${name}_$size.json")); g = d["gapped"]
    print("$envs", "|", $size, "Mbp | seed ms/step", round(d["ms_per_step"], 1), "| gapped wall calls", [round(x * 1e3, 1) for x in g["wall_s_calls"]], "GCUPS", round(g["gcups_wall"], 1), "k_ydrop", round(g["k_ydrop_ms"], 1), "ok", g.get("alignments_ok"), d["parity"].get("batch_lav_sha_ok"))
except Exception as e:
    print("$envs", $size, "failed", e, open("$O/g_${name}_$size.err").read()[-300:])
PY
  done
done
