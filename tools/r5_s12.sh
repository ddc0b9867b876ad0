#!/bin/bash
# round 5, GPU session 12: a tile's share of a partition padded to whole lines of records (LZ_PP_PAD = 8 / 16) against no padding
set -u
O=gpurun_out/r5_s12; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for k in 1 2; do STEPS=3 bash tools/ab_lib.sh $O/$k default r5_pad8 r5_pad16 2>&1 | cut -c1-250; done | tee $O/ab.txt
