"""One lastz process per GPU (BASELINE.json configs[3] / [4]): launcher and output merger for the reference CLI bound
to liblzgpu.so (oracle/_ref/lastz_gpu, integration/lzgpu_shim.c).

    python -m lastz_amd.multi --ranks N [--lastz PATH] -- <target> <query> [lastz options] > merged.lav

Every rank runs the SAME command on the SAME files.  The (query sequence, strand) units -- the granularity at
which the hot path shards exactly (src/seed_search.c:362, src/gapped_extend.c:1051) -- are dealt out by longest
processing time first (lastz_amd/shard.py) through a plan file; rank 0 builds the position table on its GPU and
lzgpu_table_share hands it to the other ranks (RCCL broadcast over xGMI; LZGPU_SHARE_TRANSPORT=file when the ranks
share one device).  A rank produces the stanzas of its own units only; this module puts them back in the
reference's order: queries in file order, + strand before - strand (src/lastz.c:1592-1691).  LAV only.
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

from lastz_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LASTZ = os.path.join(ROOT, "oracle", "_ref", "lastz_gpu")


def fasta_lengths(path):
    """lengths of the sequences of a FASTA file, in file order (one pass, no sequence kept)"""
    path = re.sub(r"\[.*\]$", "", path)
    out, n = [], None
    with open(path, "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if n is not None:
                    out.append(n)
                n = 0
            elif n is not None:
                n += len(line.rstrip(b"\r\n"))
            elif line.strip():                       # a FASTA without a header line
                n = len(line.rstrip(b"\r\n"))
    if n is not None:
        out.append(n)
    return out


def split_lav(text):
    """-> (d-stanza text + the m-stanza that closes the target, [((contig2, strand2), unit text)]);
    unit text = the s/h/a stanzas of one (query, strand).  One target sequence per run (the m-stanza, src/lastz.c:1761,
    is printed once per target, after the last query)."""
    chunks = text.split("#:lav\n")
    assert chunks[0] == "", "not a LAV file"
    head, units, trailer = None, [], ""
    for ch in chunks[1:]:
        ch = ch.replace("#:eof\n", "")
        k = ch.find("\nm {\n")
        if k >= 0:
            trailer = ch[k + 1:]; ch = ch[:k + 1]
        elif ch.startswith("m {\n"):
            trailer = ch; ch = ""
        if ch.startswith("d {"):
            head = ch
            continue
        if not ch:
            continue
        assert ch.startswith("s {"), ch[:40]
        line2 = ch.split("\n")[2]                       # the query line of the s-stanza: "name" start end strand contig
        f = line2.rsplit(" ", 2)
        units.append(((int(f[2]), int(f[1])), ch))
    return (head, trailer), units


def merge_lav(texts):
    """outputs of the ranks -> one LAV in the reference's order"""
    head, allu = None, []
    for t in texts:
        h, u = split_lav(t)
        head = head or h
        allu += u
    keys = [k for k, _ in allu]
    assert len(set(keys)) == len(keys), "a unit was produced by two ranks"
    allu.sort(key=lambda ku: ku[0])
    return "#:lav\n" + head[0] + "".join("#:lav\n" + u for _, u in allu) + head[1] + "#:eof\n"


def run(target, query, args=(), ranks=2, lastz=DEFAULT_LASTZ, devices=None, transport=None, env=None, keep=False):
    """-> (merged LAV text, [stderr of each rank], plan)"""
    lengths = fasta_lengths(query)
    plan = shard.plan_units(lengths, ranks)
    tmp = tempfile.mkdtemp(prefix="lzgpu_multi_")
    try:
        planf = os.path.join(tmp, "plan.txt")
        with open(planf, "w") as f:
            for r, units in enumerate(plan):
                for (qi, strand) in units:
                    f.write("%d %d %d\n" % (qi + 1, strand, r))
        procs = []
        for r in range(ranks):
            e = dict(os.environ)
            e.update(env or {})
            e.update({"LZGPU_RANK": str(r), "LZGPU_WORLD": str(ranks), "LZGPU_SHARE_DIR": tmp, "LZGPU_UNIT_PLAN": planf,
                      "LOCAL_RANK": str(devices[r] if devices else r)})
            if transport:
                e["LZGPU_SHARE_TRANSPORT"] = transport
            procs.append(subprocess.Popen([lastz, target, query] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e))
        outs, errs = [], []
        for r, p in enumerate(procs):
            o, e_ = p.communicate()
            if p.returncode != 0:
                for q in procs:
                    if q.poll() is None:
                        q.kill()
                raise RuntimeError("rank %d failed (rc %d): %s" % (r, p.returncode, e_.decode()[-2000:]))
            outs.append(o.decode()); errs.append(e_.decode())
        return merge_lav(outs), errs, plan
    finally:
        if not keep:
            shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--lastz", default=DEFAULT_LASTZ)
    ap.add_argument("--transport", default=None, help="file: ranks share one device (tests)")
    ap.add_argument("rest", nargs=argparse.REMAINDER, help="-- <target> <query> [lastz options]")
    a = ap.parse_args()
    rest = [x for x in a.rest if x != "--"]
    if len(rest) < 2:
        ap.error("need <target> <query>")
    merged, errs, _ = run(rest[0], rest[1], rest[2:], ranks=a.ranks, lastz=a.lastz, transport=a.transport)
    sys.stdout.write(merged)
    for r, e in enumerate(errs):
        if e.strip():
            sys.stderr.write("[rank %d] %s" % (r, e))


if __name__ == "__main__":
    main()
