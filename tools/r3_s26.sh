#!/bin/bash
# traceback window of 256 links against 64 (variant build)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s26
for lib in "" ""; do
  if [ -n "$lib" ]; then export LZGPU_LIB=$GRAFT_REPO_ROOT/lastz_amd/liblzgpu_$lib.so; else unset LZGPU_LIB; fi
  timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli > gpurun_out/s26/b.json 2> gpurun_out/s26/b.err
  L="$lib" python - <<'PY'
import json, os
d = json.loads(open("gpurun_out/s26/b.json").read().strip().splitlines()[-1]); g = d["gapped"]
print("lib", os.environ["L"] or "default", "wall", round(g["wall_s"], 4), "sbs", round(g["wall_s_strand_by_strand"], 4), "GCUPS", round(g["gcups_wall"], 1), "k_ydrop ms", round(g["k_ydrop_ms"], 1), "cycles/row", round(g["longest_dp"]["cycles_per_row"]), "traceback cycles", g["longest_dp"]["traceback_cycles"], "ok", g.get("alignments_ok"))
PY
done
