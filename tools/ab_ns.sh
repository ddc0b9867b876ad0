#!/bin/bash
# A/B of variant builds at the north star's size (200 Mbp x 200 Mbp): bash tools/ab_ns.sh <outdir> <tag|default> ...
set -u
O=$1; shift; mkdir -p $O
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for tag in "$@"; do
  lib=$PWD/lastz_amd/liblzgpu.so; [ "$tag" != "default" ] && lib=$PWD/lastz_amd/liblzgpu_$tag.so
  LZGPU_LIB=$lib timeout 600 python bench.py --north-star --steps 2 --warmup 1 --no-cpu-baseline --no-content --no-cli --no-gapped > $O/ns_$tag.json 2> $O/ns_$tag.err
  TAG=$tag O=$O python - <<'PY'
import json, os
try:
    d = json.load(open("%s/ns_%s.json" % (os.environ["O"], os.environ["TAG"])))
    d = d.get("north_star", d)
    print(os.environ["TAG"], "| 200M ms/step", round(d["ms_per_step"], 1), "frac", round(d["roofline"]["frac"], 3), {k: round(v, 1) for k, v in d["kernel_ms_per_step"].items() if v > 20}, d.get("parity", {}).get("hsp_rows"))
except Exception as e:
    print(os.environ["TAG"], "| failed", e, open("%s/ns_%s.err" % (os.environ["O"], os.environ["TAG"])).read()[-400:])
PY
done
