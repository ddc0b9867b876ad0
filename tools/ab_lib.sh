#!/bin/bash
# A/B of variant builds (lastz_amd/liblzgpu_<tag>.so, made with LZGPU_BUILD_TAG / LZGPU_CXXFLAGS) on the seed bench:
#   bash tools/ab_lib.sh <outdir> <tag|default[:ENV=VAL,...]> ...      one bench line each (kernel times, parity)
set -u
O=$1; shift; mkdir -p $O
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for spec in "$@"; do
  tag=${spec%%:*}; envs=""; [ "$spec" != "$tag" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  lib=$PWD/lastz_amd/liblzgpu.so; [ "$tag" != "default" ] && lib=$PWD/lastz_amd/liblzgpu_$tag.so
  name=$(echo "$spec" | tr ':=,' '___')
  env LZGPU_LIB=$lib $envs timeout 300 python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-cli --no-north-star --no-content ${BENCH_ARGS:---no-gapped} > $O/bench_$name.json 2> $O/bench_$name.err
  NAME=$name O=$O python - <<'PY'
import json, os
name = os.environ["NAME"]; O = os.environ["O"]
try:
    d = json.load(open("%s/bench_%s.json" % (O, name)))
    g = d.get("gapped")
    extra = (" | gapped %.1f ms %.1f GCUPS k_ydrop %.1f cyc/row %d ok %s" % (g["wall_s"] * 1e3, g["gcups_wall"], g["k_ydrop_ms"], g["longest_dp"]["cycles_per_row"], g.get("alignments_ok"))) if g else ""
    print(name, "|", round(d["ms_per_step"], 1), d["parity"]["hsp_sha_ok"], "frac", round(d["roofline"]["frac"], 3), {k: round(v, 1) for k, v in d["kernel_ms_per_step"].items() if v > 1.5}, extra)
except Exception as e:
    print(name, "| failed", e, open("%s/bench_%s.err" % (O, name)).read()[-600:])
PY
done
