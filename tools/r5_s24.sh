#!/bin/bash
# round 5, GPU session 24: the whole configs[3] job on one GPU through the N > 1 path (B3 on its thread beside the next unit's search): the launcher's rule
# (two-wave DP kernel for the big launches) against the four-wave kernel everywhere
set -u
O=gpurun_out/r5_s24; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
for v in narrow0 default; do
  E="X=1"; [ $v = narrow0 ] && E="LZGPU_DP_NARROW=0"
  env $E LZ_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 1 --force-multi --steps 1 --warmup 1 --no-cpu-baseline > $O/multi_$v.json 2> $O/multi_$v.err
  V=$v O=$O python - <<'PY'
import json,os
d=json.loads([l for l in open("%s/multi_%s.json"%(os.environ["O"],os.environ["V"])).read().split("\n") if l.startswith("{")][-1])
r=d["per_rank"][0]
print(os.environ["V"], "ms/step", round(d["ms_per_step"],1), "value", round(d["value"],4), "search_s", round(r["search_s"],2), "gapped_s", round(r["gapped_s"],2), "gcups", round(r["gapped_gcups"],1), "overlap", d["overlap"]["rank0_wall_over_sum"], "sha ok", d["north_star_unit"]["hsp_sha_ok"], "b3 batches", r.get("b3_batches"))
PY
done 2>&1 | tee $O/ab.txt
