#!/bin/bash
# configs[4]-style check at a few tens of Mbp: the lastz command line with --chain --inner=2000 --scores=HOXD70.q on
# two ranks (one GPU, file transport) against the single-process run of the same bound binary: identical LAV
cd $GRAFT_REPO_ROOT
python - <<'PY'
from lastz_amd import seqio
t, q = seqio.synth_pair(20_000_000, 24_000_000, seed=77)
seqio.write_fasta("/tmp/mt.fa", [("chrT", t)])
seqio.write_fasta("/tmp/mq.fa", [("q1", q[:7_000_000]), ("q2", q[7_000_000:12_000_000]), ("q3", q[12_000_000:20_000_000]), ("q4", q[20_000_000:])])
PY
cd /tmp
ARGS="--chain --inner=2000 --scores=$GRAFT_REPO_ROOT/lastz_amd/data/HOXD70.q --ydrop=9430"
s=$(date +%s.%N); $GRAFT_REPO_ROOT/integration/_build/lastz_gpu mt.fa mq.fa $ARGS > single.lav 2> single.err; e=$(date +%s.%N); python -c "print('single process %.2f s' % ($e - $s))"
s=$(date +%s.%N); python -m lastz_amd.multi --ranks 2 --transport file -- mt.fa mq.fa $ARGS > multi.lav 2> multi.err; rc=$?; e=$(date +%s.%N); python -c "print('two ranks %.2f s rc=$rc' % ($e - $s))"
python - <<'PY'
import sys
sys.path.insert(0, "/root/repo/tests")
from lavparse import normalize_lav
a, b = open("/tmp/single.lav").read(), open("/tmp/multi.lav").read()
print("bytes", len(a), len(b), "identical:", normalize_lav(a) == normalize_lav(b), "a-stanzas:", a.count("\na {"))
PY
tail -3 multi.err
