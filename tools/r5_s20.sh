#!/bin/bash
# round 5, GPU session 20: the walks with consecutive LDS addresses on rows whose band does not cross the end of the ring, against the build before
set -u
O=gpurun_out/r5_s20; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
STEPS=2 BENCH_ARGS=" " bash tools/ab_lib.sh $O r5_prev default r5_prev default 2>&1 | cut -c1-20,240-400 | tee $O/ab.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5_s20/bench_*.json')):
    g=json.load(open(f))['gapped']; print(f, round(g['wall_s']*1e3,1), round(g['k_ydrop_ms'],1), g['k_ydrop_builds'], 'sbs', round(g['wall_s_strand_by_strand']*1e3,1))
PY
timeout 600 python -m pytest tests/test_gpu_gapped.py -m gpu -x -q > $O/pytest_a.txt 2>&1; tail -2 $O/pytest_a.txt
