#!/bin/bash
# round 3, GPU session 1: parity of the rewritten seed kernels, A/B against round 2's settle / partition, kernel
# stats and SQ counters.  usage (through gpurun): bash tools/r3_s1.sh <tag>
set -u
TAG=${1:-s1}
O=gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_REQUIRE_GPU=1
timeout 1500 python -m pytest tests/test_gpu_seed.py -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -15 $O/pytest_gpu.txt
bash tools/ab.sh "A=default" "LZGPU_SETTLE_OLD=1" "LZGPU_PARTITION_BALLOTS=1" "LZGPU_SETTLE_OLD=1 LZGPU_PARTITION_BALLOTS=1" 2>&1 | tee $O/ab.txt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cli --no-gapped > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
cut -c1-60,200- $O/kernel_stats.csv | head -3
python - <<PY
import csv
for r in list(csv.DictReader(open("$O/kernel_stats.csv")))[:12]:
    print(r["Name"][:40], r["Calls"], r["AverageNs"], r["Percentage"])
PY
rm -rf $O/stats
bash tools/gpu_pmc.sh $TAG/pmc 20000000 20000000 2>&1 | grep "k_scan_hits\|k_settle\|k_partition\|k_fill" 
