#!/bin/bash
# kernel times of variant builds with results that may be wrong (timing-only what-ifs): bash tools/r6_kern.sh <tag> ...
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for tag in "$@"; do
lib=$PWD/lastz_amd/liblzgpu.so; [ "$tag" != "default" ] && lib=$PWD/lastz_amd/liblzgpu_$tag.so
LZGPU_LIB=$lib TAG=$tag python - <<'PY'
import os, time, numpy as np, sys
sys.path.insert(0, os.getcwd())
import bench
from lastz_amd import lzgpu, seqio
lib = lzgpu.Lib(); lib.init(0)
S = int(os.environ.get("SIZE", "50000000"))
t, q = seqio.synth_pair(S, S, seed=1000)
sub, masked, ctb = bench.scoring()
lib.table_prepare(t, lib.seed("1110100110010101111", 1), ctb)
lib.query_upload(0, q); lib.query_upload(1, seqio.revcomp(q))
def step():
    lib.table_rebuild()
    for s in (0, 1):
        try: lib.seed_hit_search(masked, slot=s)
        except Exception as e: print("search failed", e)
step()
lib.profile_enable(True); lib.profile_reset()
t0 = time.perf_counter(); K = 3
for _ in range(K): step()
dt = (time.perf_counter() - t0) / K
p = lib.profile()
print(os.environ["TAG"], "| %.1f ms/step" % (dt * 1e3), {k: round(v["ms"] / K, 1) for k, v in p.items() if v["ms"] / K > 1.5})
PY
done
