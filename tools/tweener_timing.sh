#!/bin/bash
# N3: the tweener's in-between windows (--inner=2000) as device batches against the reference's own routines, same box,
# same inputs.  usage (through gpurun): bash tools/tweener_timing.sh <tag> [tlen] [qlen]
set -u
O=gpurun_out/${1:-tw}; TL=${2:-5000000}; QL=${3:-5000000}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python - <<PY
from lastz_amd import seqio
t, q = seqio.synth_pair($TL, $QL, seed=73)
seqio.write_fasta("/tmp/tw_t.fa", [("target", t)]); seqio.write_fasta("/tmp/tw_q.fa", [("query", q)])
PY
ARGS="${4:---inner=2000}"
G=$GRAFT_REPO_ROOT/integration/_build/lastz_gpu; R=$GRAFT_REPO_ROOT/oracle/_ref/lastz
cd /tmp
run() { local name=$1; shift; local s=$(date +%s.%N); "$@" > /tmp/tw_$name.lav 2> /tmp/tw_$name.err; local e=$(date +%s.%N); python -c "print('%-34s %8.2f s wall' % ('$name', $e - $s))"; }
{
echo "pair: synthetic $TL bp x $QL bp (seed 73), lastz $ARGS"
run gpu_windows_batched          env LZGPU_VERBOSE=1 $G tw_t.fa tw_q.fa $ARGS
run gpu_windows_batched_again    env LZGPU_VERBOSE=1 $G tw_t.fa tw_q.fa $ARGS
run gpu_windows_reference_path   env LZGPU_VERBOSE=1 LZGPU_NO_WINDOW_BATCH=1 $G tw_t.fa tw_q.fa $ARGS
run gpu_no_tweener               $G tw_t.fa tw_q.fa
run pristine_lastz_1core         $R tw_t.fa tw_q.fa $ARGS
run pristine_lastz_no_tweener    $R tw_t.fa tw_q.fa
python - <<PY
a=open("/tmp/tw_gpu_windows_batched.lav").read(); b=open("/tmp/tw_gpu_windows_reference_path.lav").read(); c=open("/tmp/tw_pristine_lastz_1core.lav").read()
strip=lambda s: "\n".join(l for i,l in enumerate(s.split("\n")) if i != 2)
print("byte-identical LAV (batched == reference-path == pristine):", strip(a)==strip(b)==strip(c), " alignment blocks:", a.count("\na {"))
e=open("/tmp/tw_gpu_windows_reference_path.err").read()
print("windows per run (reference-path tables built):", e.count("[lzgpu] table: reference path"))
e=open("/tmp/tw_gpu_windows_batched.err").read()
print("batched: searched %d, extended %d batches" % (e.count("windows searched on the GPU"), e.count("windows extended on the GPU")))
PY
} | tee $GRAFT_REPO_ROOT/$O/tweener_timing.txt
