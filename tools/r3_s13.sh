#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s13
python - <<'PY'
from lastz_amd import seqio
t, q = seqio.synth_pair(50_000_000, 50_000_000, seed=1000)
seqio.write_fasta("/tmp/t.fa", [("target", t)]); seqio.write_fasta("/tmp/q.fa", [("query", q)])
PY
cd /tmp
for i in 1 2; do
  python -c "import time; print('[start] %.3f' % (time.monotonic() % 100000))"
  LZGPU_VERBOSE_CLOCK=1 $GRAFT_REPO_ROOT/integration/_build/lastz_gpu t.fa q.fa --ydrop=9430 > /tmp/out.lav 2> $GRAFT_REPO_ROOT/gpurun_out/s13/err$i.txt
  python -c "import time; print('[end] %.3f' % (time.monotonic() % 100000))"
  grep clock $GRAFT_REPO_ROOT/gpurun_out/s13/err$i.txt | cut -c1-120
done
head -c 300 /tmp/t.fa | head -3 | cut -c1-100
cd $GRAFT_REPO_ROOT; timeout 900 python -m pytest tests/test_gpu_lastz_cli.py tests/test_gpu_base_tests.py -x -q -m gpu 2>&1 | tail -3
