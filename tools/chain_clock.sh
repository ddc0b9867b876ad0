#!/bin/bash
# the host time of reduce_to_chain (src/chain.c:497) per (query, strand) inside the CLI's timeline, at the
# bench pair's size (~79 k anchors per strand) and at the north star's (~300 k): the bound binary with the reference's own
# stage clocks (-DdbgTiming), --chain, default gapped run.   bash tools/chain_clock.sh <outdir>
set -u
O=$1; mkdir -p $O
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for SZ in 50000000 200000000; do
  D=$(mktemp -d)
  python - <<PY
import sys; sys.path.insert(0, ".")
from lastz_amd import seqio
t, q = seqio.synth_pair($SZ, $SZ, seed=1000)
seqio.write_fasta("$D/t.fa", [("target", t)]); seqio.write_fasta("$D/q.fa", [("query", q)])
PY
  for FLAGS in "--ydrop=9430" "--ydrop=9430 --chain"; do
    T0=$(date +%s.%N)
    ( cd $D && timeout 900 "$GRAFT_REPO_ROOT/integration/_build/lastz_gpu_timing" t.fa q.fa $FLAGS > out.lav 2> err.txt )
    T1=$(date +%s.%N)
    echo "== $SZ bp x $SZ bp, $FLAGS: wall $(echo "$T1 - $T0" | bc) s, $(grep -c '^a {' $D/out.lav) blocks" | tee -a $O/chain_clock.txt
    grep -a "seed hit search\|chaining\|gapped extension\|total query time\|total run time\|chaining reduced" $D/err.txt | tee -a $O/chain_clock.txt
  done
  rm -rf $D
done
