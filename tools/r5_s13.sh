#!/bin/bash
# round 5, GPU session 13: what-if — the summaries of the queued scans stored densely (task order) instead of scattered
# to their hits' places (timing only: bounds what an indirection through the task list could win in k_scan_tasks)
set -u
O=gpurun_out/r5_s13; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for k in 1 2; do STEPS=3 bash tools/ab_lib.sh $O/$k default r5_tdense 2>&1 | cut -c1-250; done | tee $O/ab.txt
