#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s14
python - <<'PY'
from lastz_amd import seqio
t, q = seqio.synth_pair(50_000_000, 50_000_000, seed=1000)
seqio.write_fasta("/tmp/t.fa", [("target", t)]); seqio.write_fasta("/tmp/q.fa", [("query", q)])
PY
cd /tmp
for early in 0 1 0 1 0 1; do
  s=$(date +%s.%N)
  LZGPU_EARLY_INIT=$early LZGPU_VERBOSE_CLOCK=1 $GRAFT_REPO_ROOT/integration/_build/lastz_gpu t.fa q.fa --ydrop=9430 > /tmp/out.lav 2> $GRAFT_REPO_ROOT/gpurun_out/s14/err$early.txt
  e=$(date +%s.%N); python -c "print('early init $early: %.2f s' % ($e - $s))"
done
grep clock $GRAFT_REPO_ROOT/gpurun_out/s14/err1.txt | cut -c1-100
sha256sum /tmp/out.lav
