#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY'
from lastz_amd import seqio
t, q = seqio.synth_pair(50_000_000, 50_000_000, seed=1000)
seqio.write_fasta("/tmp/t.fa", [("target", t)]); seqio.write_fasta("/tmp/q.fa", [("query", q)])
PY
cd /tmp
for cap in 1073741824 2147483648 1073741824 2147483648 1073741824 2147483648; do
  s=$(date +%s.%N)
  LZGPU_HIT_CAPACITY=$cap $GRAFT_REPO_ROOT/integration/_build/lastz_gpu t.fa q.fa --ydrop=9430 > /tmp/out.lav 2> /tmp/err.txt
  e=$(date +%s.%N); python -c "print('cap $cap: %.2f s' % ($e - $s))"
done
sha256sum /tmp/out.lav
