#!/bin/bash
# round 5, GPU session 22: the 200 Mbp pair's gapped leg, this build against the one before walk 2 became selects (r5_prev), in one session
set -u
O=gpurun_out/r5_s22; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
for v in r5_prev default r5_prev default; do
  L=$PWD/lastz_amd/liblzgpu.so; [ $v != default ] && L=$PWD/lastz_amd/liblzgpu_$v.so
  LZGPU_LIB=$L LZGPU_DPPROF=1 timeout 600 python bench.py --north-star --steps 1 --warmup 1 --no-cpu-baseline --no-content --no-pmc --no-cli > $O/bench_$v.json 2> $O/bench_$v.err
  V=$v O=$O python - <<'PY'
import json,os
d=json.loads([l for l in open("%s/bench_%s.json"%(os.environ["O"],os.environ["V"])).read().split("\n") if l.startswith("{")][-1])
g=d.get("gapped")
print(os.environ["V"], "ms/step", round(d["ms_per_step"],1), "gapped", round(g["wall_s"]*1e3,1), "ms", round(g["gcups_wall"],1), "GCUPS kernels", round(g["k_ydrop_ms"],1), g.get("k_ydrop_builds"), g.get("alignments_ok"), "sbs", round(g["wall_s_strand_by_strand"]*1e3,1))
PY
  grep "dpprof\] launch" $O/bench_$v.err | head -3 | cut -c1-200
done 2>&1 | tee $O/ab.txt
