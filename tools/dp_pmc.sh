#!/bin/bash
# PMC passes over the gapped leg of the bench pair (k_ydrop): bash tools/dp_pmc.sh <outdir> "<counters of pass 1>" "<counters of pass 2>" ...
set -u
O=$1; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
k=0
for CNT in "$@"; do
  k=$((k+1))
  timeout 400 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $O/p$k -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-north-star --no-content > /dev/null 2> $O/err$k.txt
  python tools/pmc_agg.py "$O/p$k/**/*counter_collection.csv" | grep "k_ydrop\|^kernel\|^name" | tee $O/pmc$k.txt
  rm -rf $O/p$k
done
