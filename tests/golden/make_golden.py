"""Generate the stage-level golden vectors under tests/golden/ from the PRISTINE reference binary
(oracle/_ref/lastz, built by oracle/Makefile from /root/reference/src where the sources lie).

Run in the build container only:   python tests/golden/make_golden.py
The GPU box never runs this; it only reads the committed outputs.

Files copied verbatim from the reference's own test data (data fixtures, MIT):
  pseudocat.fa pseudopig.fa base_test.{default,hsp,hits,extended,chained}.lav
Files produced here, per case <name>:
  <name>.npz        target / query bytes (inputs)
  <name>.hsp.tsv    `--nogapped` HSPs in discovery order:
                    name2 start1(1-based) end1 start2 end2 strand2 score
  <name>.lav        default gapped run (both strands)
  <name>.stats.json collect_stats counters W,H,E,X,C (oracle/_ref/lastz_stats --stats)
With --options: options_<opt>_<pair>.lav, the --noytrim / --allgappedbounds runs (see options()).
"""
import json
import os
import re
import subprocess
import sys
import tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from lastz_amd import seqio  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "lastz")
REF_STATS = os.path.join(ROOT, "oracle", "_ref", "lastz_stats")


def adversarial(seed=3):
    """lower-case runs, N runs, tandem repeats (diagEnd collisions), planted homology"""
    rng = np.random.default_rng(seed)
    t, q = seqio.synth_pair(60000, 60000, seed=seed, block_min=500, block_max=4000)
    t = t.copy(); q = q.copy()
    unit = np.frombuffer(b"ACGTTGCAAGGCTTAACCGATCGGATCCAT", dtype=np.uint8)
    rep = np.tile(unit, 100)
    t[10000:10000 + len(rep)] = rep
    q[30000:30000 + len(rep)] = rep
    q[45000:45000 + len(rep) // 2] = rep[: len(rep) // 2]
    for arr in (t, q):
        for _ in range(12):
            s = int(rng.integers(0, len(arr) - 600)); n = int(rng.integers(20, 500))
            arr[s:s + n] |= 0x20                                   # soft-masked (lower case) run
        for _ in range(6):
            s = int(rng.integers(0, len(arr) - 300)); n = int(rng.integers(1, 200))
            arr[s:s + n] = ord("N")
    return t, q


CASES = {
    "synth200k": lambda: seqio.synth_pair(200000, 200000, seed=11),
    "synth_overlap": lambda: seqio.synth_pair(40000, 150000, seed=5, block_min=3000, block_max=12000,
                                              homolog_frac=0.8),
    "adversarial": adversarial,
}


def parse_stats(text):
    keys = {"words in seq 2": "words", "raw seed hits": "raw_hits", "GF extensions": "extensions",
            "bp extended": "bp_extended", "HSPs": "hsps", "DP cells visited": "dp_cells",
            "anchors extended": "anchors_extended", "gapped extensions": "gapped_extensions"}
    out = {}
    for line in text.split("\n"):
        m = re.match(r"\s*([^:]+):\s+([\d,]+)", line)
        if m and m.group(1).strip() in keys:
            out[keys[m.group(1).strip()]] = int(m.group(2).replace(",", ""))
    return out


def main():
    for name, gen in CASES.items():
        t, q = gen()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), target=t, query=q)
        with tempfile.TemporaryDirectory() as d:
            tf, qf = os.path.join(d, "t.fa"), os.path.join(d, "q.fa")
            seqio.write_fasta(tf, [("target", t)])
            seqio.write_fasta(qf, [("query", q)])
            hsp = subprocess.check_output([REF, tf, qf, "--nogapped",
                                           "--format=general-:name2,start1,end1,start2,end2,strand2,score"])
            open(os.path.join(HERE, name + ".hsp.tsv"), "wb").write(hsp)
            lav = subprocess.check_output([REF, tf, qf]).decode().replace(d + "/", "")
            open(os.path.join(HERE, name + ".lav"), "w").write(lav)
            st = os.path.join(d, "st.txt")
            subprocess.check_output([REF_STATS, tf, qf, "--stats=" + st], stderr=subprocess.DEVNULL)
            json.dump(parse_stats(open(st).read()), open(os.path.join(HERE, name + ".stats.json"), "w"), indent=1)
        print(name, len(hsp.split(b"\n")) - 1, "HSPs;", lav.count("a {"), "gapped blocks")


def options():
    """--noytrim / --allgappedbounds runs of the pristine binary on the pairs of tests/test_oracle_vs_reference.py::
    _option_pairs (inputs are rebuilt from the committed cases / seeds there): options_<opt>_<pair>.lav"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_oracle_vs_reference as T
    for opt, (flags, _) in T.OPTION_CASES.items():
        for name, (t, q) in T._option_pairs().items():
            with tempfile.TemporaryDirectory() as d:
                tf, qf = os.path.join(d, "t.fa"), os.path.join(d, "q.fa")
                seqio.write_fasta(tf, [("target", t)]); seqio.write_fasta(qf, [("query", q)])
                lav = subprocess.check_output([REF, tf, qf] + flags).decode().replace(d + "/", "")
                open(os.path.join(HERE, "options_%s_%s.lav" % (opt, name)), "w").write(lav)
                print(opt, name, lav.count("a {"), "gapped blocks")


if __name__ == "__main__":
    options() if "--options" in sys.argv else main()
