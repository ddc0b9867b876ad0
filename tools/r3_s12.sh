#!/bin/bash
# session 12: where the CLI's first search segment goes -- device buffer (re)allocations, the reference's own query I/O
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s12
python - <<'PY'
from lastz_amd import seqio
t, q = seqio.synth_pair(50_000_000, 50_000_000, seed=1000)
seqio.write_fasta("/tmp/t.fa", [("target", t)]); seqio.write_fasta("/tmp/q.fa", [("query", q)])
seqio.write_fasta("/tmp/ts.fa", [("target", t[:20000])])
PY
cd /tmp
for i in 1 2; do
  LZGPU_HOSTPROF=1 LZGPU_VERBOSE_CLOCK=1 LZGPU_VERBOSE=1 $GRAFT_REPO_ROOT/integration/_build/lastz_gpu t.fa q.fa --ydrop=9430 > /tmp/out.lav 2> $GRAFT_REPO_ROOT/gpurun_out/s12/err$i.txt
done
# the reference's own cost of reading the query (both strands) and the target: tiny counterpart
s=$(date +%s.%N); $GRAFT_REPO_ROOT/oracle/_ref/lastz ts.fa q.fa --nogapped > /dev/null; e=$(date +%s.%N); python -c "print('pristine: 20 kbp target x 50 Mbp query (query I/O + 2 strands of search on a tiny table) %.2f s' % ($e - $s))" | tee $GRAFT_REPO_ROOT/gpurun_out/s12/io.txt
s=$(date +%s.%N); $GRAFT_REPO_ROOT/oracle/_ref/lastz t.fa ts.fa --nogapped > /dev/null; e=$(date +%s.%N); python -c "print('pristine: 50 Mbp target x 20 kbp query (target I/O + table build on the CPU) %.2f s' % ($e - $s))" | tee -a $GRAFT_REPO_ROOT/gpurun_out/s12/io.txt
s=$(date +%s.%N); $GRAFT_REPO_ROOT/oracle/_ref/lastz ts.fa q.fa --nogapped --strand=plus > /dev/null; e=$(date +%s.%N); python -c "print('pristine: same, plus strand only %.2f s' % ($e - $s))" | tee -a $GRAFT_REPO_ROOT/gpurun_out/s12/io.txt
grep -c . $GRAFT_REPO_ROOT/gpurun_out/s12/err2.txt
grep "device buffer\|clock" $GRAFT_REPO_ROOT/gpurun_out/s12/err2.txt | cut -c1-160
cd $GRAFT_REPO_ROOT; timeout 600 python -m pytest tests/test_gpu_multi.py -x -q -m gpu -k "line_oriented" 2>&1 | tail -3
