#!/bin/bash
# round 4, GPU session 3: phase-copy planar layout of the 2-bit codes (k_scan_hits), gapped host profile
set -u
O=gpurun_out/r4_s3; mkdir -p $O
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
export LZGPU_REQUIRE_GPU=1
timeout 900 python -m pytest tests/test_gpu_seed.py -m gpu -x -q > $O/pytest_seed.txt 2>&1; tail -5 $O/pytest_seed.txt
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cli --no-gapped"
run() { tag=$1; shift; env "$@" timeout 300 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; TAG=$tag O=$O python - <<'PY'
import json, os
tag = os.environ["TAG"]; O = os.environ["O"]
try:
    d = json.load(open("%s/bench_%s.json" % (O, tag)))
    print(tag, "|", round(d["ms_per_step"], 1), d["parity"]["hsp_sha_ok"], "frac", round(d["roofline"]["frac"], 3), {k: round(v, 1) for k, v in d["kernel_ms_per_step"].items() if v > 1.5})
except Exception as e:
    print(tag, "| failed", e, open("%s/bench_%s.err" % (O, tag)).read()[-600:])
PY
}
run tpb640 A=1
run tpb768 LZGPU_SC_TPB=768
run tpb512 LZGPU_SC_TPB=512
run tpb640b A=1
LZGPU_HOSTPROF=1 timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cli > $O/bench_g.json 2> $O/bench_g.err
grep -a "hostprof\] gapped" $O/bench_g.err | tail -4
python - <<PY
import json
d = json.load(open("$O/bench_g.json")); g = d["gapped"]
print("ms/step", round(d["ms_per_step"],1), "| gapped wall", round(g["wall_s"]*1e3,1), "ms  strand-by-strand", round(g["wall_s_strand_by_strand"]*1e3,1), "GCUPS", round(g["gcups_wall"],1), "k_ydrop", round(g["k_ydrop_ms"],1), g["k_ydrop_launches"], "cyc/row", round(g["longest_dp"]["cycles_per_row"]), "ok", g.get("alignments_ok"))
PY
