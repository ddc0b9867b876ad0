#!/bin/bash
# host profile + bench numbers of the gapped leg: bash tools/gapped_prof.sh <outdir> [runs]
set -u
O=$1; N=${2:-2}; mkdir -p $O
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for k in $(seq 1 $N); do
LZGPU_HOSTPROF=1 timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cli --no-north-star --no-content > $O/bench_$k.json 2> $O/bench_$k.err
grep -a "hostprof\] gapped" $O/bench_$k.err | tail -3
python - <<PY
import json
d = json.load(open("$O/bench_$k.json")); g = d["gapped"]
print("ms/step", round(d["ms_per_step"],1), "| gapped wall", round(g["wall_s"]*1e3,1), "ms  strand-by-strand", round(g["wall_s_strand_by_strand"]*1e3,1), "GCUPS", round(g["gcups_wall"],1), "k_ydrop", round(g["k_ydrop_ms"],1), g["k_ydrop_launches"], "cyc/row", round(g["longest_dp"]["cycles_per_row"]), "ok", g.get("alignments_ok"))
PY
done
