"""SHA-256 fingerprints of the pristine reference's output on the EXACT bench.py pair (BASELINE.json configs[1]
and [2]: synthetic 50 Mbp x 50 Mbp, seed 1000): ~25 minutes of one CPU core per run, build container only.

  python tests/golden/make_bench_sha.py            -> tests/golden/bench50m.sha.json

  hsp_sha : sha256 of `lastz T Q --nogapped --format=general-:name2,start1,end1,start2,end2,strand2,score`
  lav_sha : sha256 of `lastz T Q --ydrop=9430` (LAV) without the first line of the d-stanza (it echoes the
            command line)
bench.py and tests/test_gpu_lastz_cli.py recompute both from the HIP path on the GPU box and compare.
"""
import hashlib
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from lastz_amd import seqio  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "lastz")
HSP_FMT = "--format=general-:name2,start1,end1,start2,end2,strand2,score"


def lav_fingerprint(text):
    """drop the first line inside the d {} stanza (the echoed command line)"""
    lines = text.split("\n")
    for i, ln in enumerate(lines):
        if ln.startswith("d {"):
            del lines[i + 1]
            break
    return hashlib.sha256("\n".join(lines).encode()).hexdigest()


def main(tlen=50_000_000, qlen=50_000_000, seed=1000, workdir="/tmp/bench50m"):
    os.makedirs(workdir, exist_ok=True)
    tf, qf = os.path.join(workdir, "t.fa"), os.path.join(workdir, "q.fa")
    if not (os.path.exists(tf) and os.path.exists(qf)):
        t, q = seqio.synth_pair(tlen, qlen, seed=seed)
        seqio.write_fasta(tf, [("target", t)]); seqio.write_fasta(qf, [("query", q)])
    t0 = time.time()
    # (the LAV s-stanzas quote the file names as given: relative names, run from the directory)
    p1 = subprocess.Popen([REF, "t.fa", "q.fa", "--nogapped", HSP_FMT], stdout=open(os.path.join(workdir, "hsp.tsv"), "wb"), cwd=workdir)
    p2 = subprocess.Popen([REF, "t.fa", "q.fa", "--ydrop=9430"], stdout=open(os.path.join(workdir, "gapped.lav"), "wb"), cwd=workdir)
    assert p1.wait() == 0
    t1 = time.time() - t0
    assert p2.wait() == 0
    t2 = time.time() - t0
    hsp = open(os.path.join(workdir, "hsp.tsv"), "rb").read()
    lav = open(os.path.join(workdir, "gapped.lav")).read()
    out = {"tlen": tlen, "qlen": qlen, "seed": seed,
           "hsp_rows": hsp.count(b"\n"), "hsp_sha": hashlib.sha256(hsp).hexdigest(),
           "lav_blocks": lav.count("\na {"), "lav_sha": lav_fingerprint(lav),
           "reference_wall_s": {"nogapped": round(t1, 1), "gapped": round(t2, 1)}, "reference": "lastz 1.04.58, 1 core each"}
    json.dump(out, open(os.path.join(HERE, "bench50m.sha.json"), "w"), indent=1)
    print(out)


if __name__ == "__main__":
    main()
