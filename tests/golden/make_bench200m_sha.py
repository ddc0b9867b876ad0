"""SHA-256 fingerprints of the pristine reference's output on bench.py's NORTH-STAR pair (BASELINE.json north_star:
synthetic 200 Mbp x 200 Mbp, seed 1000), build container only.

One reference process would take ~10.5 h on this pair; the (sequence, strand) units of a lastz run are independent
(SURVEY.md 8e), so the two strands run as two processes (`--strand=plus`, `--strand=minus`), for both command lines:
four processes, ~5.5 h of wall clock on four cores.  The whole run's output is the + strand's followed by the - strand's
(general format: plain concatenation; LAV: the + run without its closing `m {}` stanza and `#:eof`, then the - run from its first `s {` stanza on);
`--check` proves that assembly byte for byte on a 1 Mbp pair against a single two-strand process.

  python tests/golden/make_bench200m_sha.py --check
  python tests/golden/make_bench200m_sha.py            -> tests/golden/bench200m.sha.json   (hours)

  hsp_sha : sha256 of `lastz T Q --nogapped --format=general-:name2,start1,end1,start2,end2,strand2,score`
  lav_sha : sha256 of `lastz T Q --ydrop=9430` (LAV) without the first line of the d-stanza (it echoes the command line)
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
    lines = text.split("\n")
    for i, ln in enumerate(lines):
        if ln.startswith("d {"):
            del lines[i + 1]
            break
    return hashlib.sha256("\n".join(lines).encode()).hexdigest()


def join_lav(plus, minus):
    """the two one-strand LAVs as the LAV of the two-strand run (up to the echoed command line)"""
    tail = "m {\n  n 0\n}\n#:eof\n"                            # (a run ends with its m-stanza: the + run's goes)
    assert plus.endswith(tail)
    k = minus.index("#:lav\ns {")                              # the - run's first s-stanza (its d-stanza goes)
    return plus[:-len(tail)] + minus[k:]


def run_split(workdir, wait=True):
    """four processes; returns {name: Popen}"""
    procs = {}
    for name, args in (("hsp_plus", ["--nogapped", HSP_FMT, "--strand=plus"]), ("hsp_minus", ["--nogapped", HSP_FMT, "--strand=minus"]),
                       ("lav_plus", ["--ydrop=9430", "--strand=plus"]), ("lav_minus", ["--ydrop=9430", "--strand=minus"])):
        procs[name] = subprocess.Popen([REF, "t.fa", "q.fa"] + args, stdout=open(os.path.join(workdir, name + ".out"), "wb"), cwd=workdir)
    return procs


def assemble(workdir):
    rd = lambda n: open(os.path.join(workdir, n + ".out")).read()
    hsp = rd("hsp_plus") + rd("hsp_minus")
    lav = join_lav(rd("lav_plus"), rd("lav_minus"))
    return hsp, lav


def make_inputs(workdir, tlen, qlen, seed):
    os.makedirs(workdir, exist_ok=True)
    tf, qf = os.path.join(workdir, "t.fa"), os.path.join(workdir, "q.fa")
    if not (os.path.exists(tf) and os.path.exists(qf)):
        t, q = seqio.synth_pair(tlen, qlen, seed=seed)
        seqio.write_fasta(tf, [("target", t)]); seqio.write_fasta(qf, [("query", q)])


def check(workdir="/tmp/bench200m_check"):
    make_inputs(workdir, 1_000_000, 1_000_000, 1000)
    for p in run_split(workdir).values():
        assert p.wait() == 0
    hsp, lav = assemble(workdir)
    one = lambda a: subprocess.run([REF, "t.fa", "q.fa"] + a, capture_output=True, text=True, cwd=workdir, check=True).stdout
    assert hsp == one(["--nogapped", HSP_FMT]), "HSP rows of the two strands do not add up to the whole run's"
    assert lav_fingerprint(lav) == lav_fingerprint(one(["--ydrop=9430"])), "LAV of the two strands does not add up to the whole run's"
    print("assembly of the strand-split runs == the two-strand run (1 Mbp pair): ok")


def main(tlen=200_000_000, qlen=200_000_000, seed=1000, workdir="/tmp/bench200m"):
    make_inputs(workdir, tlen, qlen, seed)
    t0 = time.time()
    procs = run_split(workdir)
    walls = {}
    for name, p in procs.items():
        assert p.wait() == 0, name
        walls[name] = round(time.time() - t0, 1)
    hsp, lav = assemble(workdir)
    out = {"tlen": tlen, "qlen": qlen, "seed": seed,
           "hsp_rows": hsp.count("\n"), "hsp_sha": hashlib.sha256(hsp.encode()).hexdigest(),
           "lav_blocks": lav.count("\na {"), "lav_sha": lav_fingerprint(lav),
           "reference_wall_s_per_strand_process": walls,
           "reference": "lastz 1.04.58, four 1-core processes (two command lines x two strands), outputs joined as tests/golden/make_bench200m_sha.py --check proves",
           "note": "file names as given on the command line: run from the directory that holds t.fa / q.fa"}
    json.dump(out, open(os.path.join(HERE, "bench200m.sha.json"), "w"), indent=1)
    print(out)


if __name__ == "__main__":
    check() if "--check" in sys.argv else main()
