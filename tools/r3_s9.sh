#!/bin/bash
set -u
O=gpurun_out/${1:-s9}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
L=$GRAFT_REPO_ROOT/lastz_amd
timeout 1500 python -m pytest tests/test_gpu_gapped.py -m gpu -x -q 2>&1 | tail -4
for cfg in "A=wpe6" "LZGPU_LIB=$L/liblzgpu_wpe5.so" "LZGPU_LIB=$L/liblzgpu_wpe4.so"; do
env $cfg timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli > $O/b.json 2> $O/b.err; CFG="$cfg" python - <<PY
import json, os
try:
    d=json.load(open("$O/b.json")); g=d["gapped"]
    print(os.environ["CFG"], "| gapped wall %.4f seq %.4f gcups %.1f k_ydrop %.1f ms / %d launches ok=%s longest %s" % (g["wall_s"], g["wall_s_strand_by_strand"], g["gcups_wall"], g["k_ydrop_ms"], g["k_ydrop_launches"], g["alignments_ok"], g["longest_dp"]["cycles_per_row"]))
except Exception as e: print("failed", e, open("$O/b.err").read()[-600:])
PY
done
