#!/bin/bash
# where the wall time of the lastz CLI (GPU-bound binary) goes on the bench pair: host phases of the library
# (LZGPU_HOSTPROF) and the shim's notes (LZGPU_VERBOSE), two runs (the second has the files in the page cache)
cd $GRAFT_REPO_ROOT
python - <<'PY'
from lastz_amd import seqio
t, q = seqio.synth_pair(50_000_000, 50_000_000, seed=1000)
seqio.write_fasta("/tmp/t.fa", [("target", t)]); seqio.write_fasta("/tmp/q.fa", [("query", q)])
PY
cd /tmp
for i in 1 2; do
  s=$(date +%s.%N)
  python -c "import time; print('[start] %.3f' % (time.monotonic() % 100000))"; LZGPU_HOSTPROF=1 LZGPU_VERBOSE_CLOCK=1 LZGPU_VERBOSE=1 $GRAFT_REPO_ROOT/integration/_build/lastz_gpu t.fa q.fa --ydrop=9430 > /tmp/out.lav 2> /tmp/err.txt
  e=$(date +%s.%N); python -c "print('run $i wall %.2f s' % ($e - $s))"
  grep "clock\|hostprof" /tmp/err.txt | grep -v "device buffer" | cut -c1-200; python -c "import time; print('[end] %.3f' % (time.monotonic() % 100000))"
done
