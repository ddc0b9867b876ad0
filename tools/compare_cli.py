#!/usr/bin/env python3
"""Same box, same inputs, same flags: the pristine reference binary (oracle/_ref/lastz) vs the
GPU-bound binary (integration/_build/lastz_gpu) through the lastz CLI; byte-compares the LAV (modulo the
d-stanza's command line) and reports both wall clocks.  One-off evidence tool (not a test):
    python tools/compare_cli.py --tlen 10000000 --qlen 10000000 [--nogapped]"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from lastz_amd import seqio          # noqa: E402
from lavparse import normalize_lav   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tlen", type=int, default=10_000_000)
    ap.add_argument("--qlen", type=int, default=10_000_000)
    ap.add_argument("--seed", type=int, default=1000)
    ap.add_argument("--nogapped", action="store_true")
    ap.add_argument("--gpu-only", action="store_true", help="run only lastz_gpu; report wall, LAV size and sha256 to compare with an earlier full run")
    a = ap.parse_args()
    t, q = seqio.synth_pair(a.tlen, a.qlen, seed=a.seed)
    flags = ["--nogapped"] if a.nogapped else ["--ydrop=9430"]
    with tempfile.TemporaryDirectory() as d:
        seqio.write_fasta(os.path.join(d, "t.fa"), [("target", t)])
        seqio.write_fasta(os.path.join(d, "q.fa"), [("query", q)])
        res = {}
        outs = {}
        for name in (("lastz_gpu",) if a.gpu_only else ("lastz_gpu", "lastz")):
            t0 = time.time()
            p = subprocess.run([os.path.join(ROOT, "integration", "_build", name) if name == "lastz_gpu" else os.path.join(ROOT, "oracle", "_ref", name), "t.fa", "q.fa"] + flags, cwd=d,
                               capture_output=True, text=True)
            res[name + "_wall_s"] = round(time.time() - t0, 3)
            if p.returncode != 0:
                print(p.stderr[-2000:]); sys.exit(1)
            outs[name] = normalize_lav(p.stdout)
        import hashlib
        res["lav_sha256"] = {k: hashlib.sha256(v.encode()).hexdigest() for k, v in outs.items()}
        if a.gpu_only:
            res.update({"tlen": a.tlen, "qlen": a.qlen, "flags": flags, "lav_bytes": len(outs["lastz_gpu"]),
                        "alignment_blocks": outs["lastz_gpu"].count("a {")})
            print(json.dumps(res)); return
        res.update({"tlen": a.tlen, "qlen": a.qlen, "flags": flags, "lav_bytes": len(outs["lastz"]),
                    "alignment_blocks": outs["lastz"].count("a {"), "byte_identical": outs["lastz"] == outs["lastz_gpu"],
                    "speedup": round(res["lastz_wall_s"] / res["lastz_gpu_wall_s"], 1),
                    "note": "whole lastz CLI incl. FASTA parsing, table, both strands, output; 1 CPU core vs 1 MI355X"})
        print(json.dumps(res))


if __name__ == "__main__":
    main()
