#!/bin/bash
# round 5, GPU session 10: what the fill kernel's slow mode (27-29 ms against 18-20) depends on: its partition-byte stores (none / plain),
# all its stores plain instead of non-temporal -- each variant several times, alternating with the default
set -u
O=gpurun_out/r5_s10; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for k in 1 2 3; do STEPS=3 bash tools/ab_lib.sh $O/$k default r5_nobinst r5_fillplain r5_binsplain 2>&1 | cut -c1-260; done | tee $O/ab.txt
