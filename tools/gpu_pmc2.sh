#!/bin/bash
# one PMC pass with the counters given on the command line (space-separated string), small bench, per-kernel sums
# usage: bash tools/gpu_pmc2.sh <tag> "<counters>" [tlen] [qlen]
set -u
TAG=${1:-pmc}; CNT=$2; TL=${3:-20000000}; QL=${4:-20000000}
O=gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export LZGPU_SERIAL=1
timeout 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $O/p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-gapped --tlen $TL --qlen $QL > /dev/null 2> $O/err.txt
python tools/pmc_agg.py "$O/p/**/*counter_collection.csv" | grep -v "rocprim\|__amd" | tee $O/pmc.txt | grep "scan_hits\|k_partition\|scan_tasks\|k_settle\|k_fill\|k_hist "
rm -rf $O/p
