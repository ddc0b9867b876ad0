#!/bin/bash
# north-star size (200 Mbp x 200 Mbp) on one GPU: bench line + rocprofv3 kernel stats of the same command
set -u
O=gpurun_out/${1:-ns}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 1500 python bench.py --north-star --steps 2 --warmup 1 --cpu-sample 5000000 > $O/bench_ns.json 2> $O/bench_ns.err; tail -c 1800 $O/bench_ns.json; tail -3 $O/bench_ns.err
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --north-star --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_ns_under_rocprof.json 2> $O/rocprof.err
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_ns.csv \;
python - <<PY
import csv
for r in list(csv.DictReader(open("$O/kernel_stats_ns.csv")))[:10]:
    print(r["Name"][:40], r["Calls"], r["AverageNs"], r["Percentage"])
PY
rm -rf $O/stats
