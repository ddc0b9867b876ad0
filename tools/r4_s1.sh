#!/bin/bash
# round 4, GPU session 1: (a) instruction rates, (b) seed parity tests with the new fill kernel, (c) A/B: new fill vs the
# shuffle fill, LZ_F2_CAP, doubled query loads, (d) where k_settle2's tile time goes, (e) host profile of the gapped leg
set -u
O=gpurun_out/r4_s1; mkdir -p $O
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
export LZGPU_REQUIRE_GPU=1
timeout 120 tools/ub/valu_rate > $O/valu_rate.txt 2>&1; cat $O/valu_rate.txt
timeout 900 python -m pytest tests/test_gpu_seed.py tests/test_gpu_gapped.py -m gpu -x -q > $O/pytest_seed_gapped.txt 2>&1; tail -5 $O/pytest_seed_gapped.txt
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cli --no-gapped"
run() { tag=$1; shift; env "$@" timeout 300 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; TAG=$tag python - <<'PY'
import json, os
tag = os.environ["TAG"]
try:
    d = json.load(open("gpurun_out/r4_s1/bench_%s.json" % tag))
    print(tag, "|", round(d["ms_per_step"], 1), d["parity"]["hsp_sha_ok"], {k: round(v, 1) for k, v in d["kernel_ms_per_step"].items() if v > 1.5})
except Exception as e:
    print(tag, "| failed", e, open("gpurun_out/r4_s1/bench_%s.err" % tag).read()[-600:])
PY
}
run new_fill A=1
run shuffle_fill LZGPU_FILL_SHUFFLE=1
run cap4096 LZGPU_LIB=$PWD/lastz_amd/liblzgpu_cap4096.so
run double_query LZGPU_LIB=$PWD/lastz_amd/liblzgpu_dq.so
run new_fill_again A=1
# phase clocks (settle walker / sorter; DP phases) + host profile of the gapped leg
LZGPU_LIB=$PWD/lastz_amd/liblzgpu_clk.so LZGPU_HOSTPROF=1 LZGPU_DPPROF=1 timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cli > $O/bench_clk.json 2> $O/bench_clk.err
grep -a "phase clocks\|hostprof\] gapped\|dpprof" $O/bench_clk.err | tail -40
LZGPU_HOSTPROF=1 timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cli > $O/bench_hostprof.json 2> $O/bench_hostprof.err
grep -a "hostprof" $O/bench_hostprof.err | grep -v "device buffer" | tail -40
