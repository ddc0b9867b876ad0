#!/bin/bash
# round 5, GPU session 21: 200 Mbp pair (launches of ~9000 DPs on the two-wave kernel): one leading wave per DP (the rule's choice beyond 5632 DPs) against a copy of
# the row set-up in both waves
set -u
O=gpurun_out/r5_s21; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
for v in default repl1; do
  E=""; [ $v = repl1 ] && E="LZGPU_DP_REPL=1"
  env $E timeout 600 python bench.py --north-star --steps 1 --warmup 1 --no-cpu-baseline --no-content --no-pmc --no-cli > $O/bench_$v.json 2> $O/bench_$v.err
  V=$v O=$O python - <<'PY'
import json,os
d=json.loads([l for l in open("%s/bench_%s.json"%(os.environ["O"],os.environ["V"])).read().split("\n") if l.startswith("{")][-1])
g=d.get("gapped") or d.get("north_star",{}).get("gapped")
print(os.environ["V"], "ms/step", round(d["ms_per_step"],1), "gapped", round(g["wall_s"]*1e3,1), "ms", round(g["gcups_wall"],1), "GCUPS kernels", round(g["k_ydrop_ms"],1), g.get("k_ydrop_builds"), g.get("alignments_ok"), d.get("parity"))
PY
done 2>&1 | tee $O/ab.txt
