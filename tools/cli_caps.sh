#!/bin/bash
# wall time of the bound lastz on the bench pair against the chunk capacity (the device buffers a short-lived process must allocate):
#   bash tools/cli_caps.sh <outdir> [log2 capacities...]
O=$1; shift; mkdir -p $O
cd $GRAFT_REPO_ROOT
python - <<'PY'
from lastz_amd import seqio
t, q = seqio.synth_pair(50_000_000, 50_000_000, seed=1000)
seqio.write_fasta("/tmp/t.fa", [("target", t)]); seqio.write_fasta("/tmp/q.fa", [("query", q)])
PY
cd /tmp
$GRAFT_REPO_ROOT/integration/_build/lastz_gpu t.fa q.fa --ydrop=9430 > /tmp/out0.lav 2>/dev/null     # (page cache, first-touch of the box)
for lg in "$@"; do
  cap=$((1 << lg))
  for i in 1 2 3; do
    s=$(date +%s.%N)
    LZGPU_HIT_CAPACITY=$cap LZGPU_HOSTPROF=1 $GRAFT_REPO_ROOT/integration/_build/lastz_gpu t.fa q.fa --ydrop=9430 > /tmp/out.lav 2> /tmp/err.txt
    e=$(date +%s.%N)
    cmp -s /tmp/out.lav /tmp/out0.lav && same=same || same=DIFFERENT
    python -c "print('capacity 2^$lg run $i: wall %.3f s, output $same, hipMalloc %.0f ms in %d calls' % ($e - $s, sum(float(l.split('hipMalloc')[1].split('ms')[0]) for l in open('/tmp/err.txt', errors='replace') if 'device buffer' in l), sum(1 for l in open('/tmp/err.txt', errors='replace') if 'device buffer' in l)))"
  done
done | tee $O/cli_caps.txt
