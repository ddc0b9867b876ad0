#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY'
from lastz_amd import seqio
t, q = seqio.synth_pair(50_000_000, 50_000_000, seed=1000)
seqio.write_fasta("/tmp/t.fa", [("target", t)]); seqio.write_fasta("/tmp/q.fa", [("query", q)])
PY
cd /tmp
for i in 1 2 3 4 5 6 7 8; do
  s=$(date +%s.%N)
  LZGPU_VERBOSE_CLOCK=1 LZGPU_HOSTPROF=1 $GRAFT_REPO_ROOT/integration/_build/lastz_gpu t.fa q.fa --ydrop=9430 > /tmp/out.lav 2> /tmp/err$i.txt
  e=$(date +%s.%N); python -c "print('run $i: %.2f s' % ($e - $s))"
done
for i in 1 2 3 4 5 6 7 8; do grep -h "device buffer\|copy candidates\|count+scan\|wait for GPU" /tmp/err$i.txt | awk -v r=$i '{ if ($0 ~ /hipMalloc/) { split($0, a, "hipMalloc "); if (a[2]+0 > 20) print "run " r ": " $0 } else if ($0 ~ /total/) { n=split($0, b, " "); if (b[n-2]+0 > 300) print "run " r ": " $0 } }'; done | head -20
