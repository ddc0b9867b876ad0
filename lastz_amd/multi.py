"""One lastz process per GPU (BASELINE.json configs[3] / [4]): launcher and output merger for the reference CLI bound
to liblzgpu.so (integration/_build/lastz_gpu, integration/lzgpu_shim.c).

    python -m lastz_amd.multi --ranks N [--lastz PATH] -- <target> <query> [lastz options] > merged.lav

The (query sequence, strand) units -- the granularity at which the hot path shards exactly (src/seed_search.c:362,
src/gapped_extend.c:1051) -- are dealt out by longest processing time first (lastz_amd/shard.py) through a plan
file; rank 0 builds the position table on its GPU and lzgpu_table_share hands it to the other ranks (RCCL broadcast
over xGMI; LZGPU_SHARE_TRANSPORT=file when the ranks share one device).

A rank pays host time only for the query sequences it owns: the launcher indexes the query FASTA once (offsets of
the records) and gives every rank its own query file in which the records of the other ranks are reduced to
their header lines.  The reference skips an empty record with a warning (src/lastz.c:1478-1484) after counting it
(src/sequences.c:2079), so the contig numbers -- which LAV prints -- stay those of the original file, and a rank
neither parses nor reverse-complements a base it does not own.  (Round 2 ran the whole host loop of every query on
every rank: about 1 s of host time per 50 Mbp of query, whoever owned it.)  Queries that are not plain FASTA files
(2bit, hsx, files with actions other than the ones passed through) fall back to "every rank reads everything".

A rank produces the stanzas of its own units only; this module puts them back in the reference's order: queries in
file order, + strand before - strand (src/lastz.c:1592-1691).  LAV is merged by its stanzas; MAF, AXT, general, cigar
and differences by unit markers the bound binary prints (LZGPU_UNIT_MARKERS); anything else is refused before
anything is launched.
"""
import argparse
import mmap
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

from lastz_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LASTZ = os.path.join(ROOT, "integration", "_build", "lastz_gpu")

_ACTIONS = re.compile(r"((?:\[[^\]]*\])*)$")


def split_spec(spec):
    """'path[action][action]' -> (path, '[action][action]')"""
    m = _ACTIONS.search(spec)
    return spec[:m.start()], m.group(1)


def fasta_index(path):
    """records of a FASTA file in file order: [(header offset, body offset, end offset, bases)] -- one pass over a
    memory map, no sequence kept.  `bases` counts the non-newline bytes of the body (what the unit plan weighs).
    A file without a header line is one record."""
    path = split_spec(path)[0]
    size = os.path.getsize(path)
    if size == 0:
        return []
    out = []
    with open(path, "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as m:
        starts = []
        if m[0:1] == b">":
            starts.append(0)
        k = m.find(b"\n>")
        while k >= 0:
            starts.append(k + 1)
            k = m.find(b"\n>", k + 1)
        if not starts or starts[0] != 0:
            first_end = starts[0] if starts else size
            if m[0:first_end].strip():
                out.append((0, 0, first_end, _count_bases(m, 0, first_end)))   # leading sequence without a header
        for i, s in enumerate(starts):
            e = starts[i + 1] if i + 1 < len(starts) else size
            nl = m.find(b"\n", s, e)
            body = e if nl < 0 else nl + 1
            out.append((s, body, e, _count_bases(m, body, e)))
    return out


def _count_bases(m, a, b):
    n, step = 0, 1 << 24
    for lo in range(a, b, step):
        chunk = m[lo:min(b, lo + step)]
        n += len(chunk) - chunk.count(b"\n") - chunk.count(b"\r")
    return n


def fasta_lengths(path):
    """lengths of the sequences of a FASTA file, in file order"""
    return [r[3] for r in fasta_index(path)]


def plan_units(lengths, ranks, whole_sequences=None):
    """(query, strand) units -> ranks.  Whole sequences (both strands on one rank: the rank parses and reverse-
    complements the sequence once) whenever that costs at most 5 % in the longest rank's load against dealing the
    strands out separately; else strand by strand.  whole_sequences = True / False forces one or the other."""
    by_unit = shard.plan_units(lengths, ranks)
    by_seq = [[(i, s) for (i, _) in p for s in (0, 1)] for p in shard.plan_units(lengths, ranks, strands=1)]
    load = lambda plan: max((sum(lengths[i] for i, _ in p) for p in plan), default=0)
    if whole_sequences is None:
        whole_sequences = load(by_seq) <= 1.05 * load(by_unit)
    return by_seq if whole_sequences else by_unit


def write_rank_query(src, index, owned, dst):
    """the query file of one rank: every record's header line, the body of the owned records only"""
    with open(src, "rb") as fi, open(dst, "wb") as fo:
        for qi, (hoff, boff, eoff, _) in enumerate(index):
            lo, hi = (hoff, eoff) if qi in owned else (hoff, boff)
            fi.seek(lo)
            left, last = hi - lo, b"\n"
            while left > 0:
                buf = fi.read(min(left, 1 << 24))
                if not buf:
                    break
                fo.write(buf)
                left -= len(buf)
                last = buf[-1:]
            if last != b"\n":
                fo.write(b"\n")


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


def merge_lav(texts, rename=None):
    """outputs of the ranks -> one LAV in the reference's order.  rename = [(rank's query file, original query file)]:
    the s-stanza names the query FILE (src/lav.c:111-149); a rank that read its own copy prints that copy's path."""
    head, allu = None, []
    for r, t in enumerate(texts):
        h, u = split_lav(t)
        if rename and rename[r][0] != rename[r][1]:
            u = [(k, _rename_query(x, rename[r][0], rename[r][1])) for k, x in u]
            h = (h[0].replace(" " + rename[r][0], " " + rename[r][1], 1) if h[0] else h[0], h[1])   # the command line in the d-stanza
        head = head or h
        allu += u
    keys = [k for k, _ in allu]
    assert len(set(keys)) == len(keys), "a unit was produced by two ranks"
    allu.sort(key=lambda ku: ku[0])
    return "#:lav\n" + head[0] + "".join("#:lav\n" + u for _, u in allu) + head[1] + "#:eof\n"


def _rename_query(unit, old, new):
    lines = unit.split("\n", 3)                          # "s {", target line, query line, rest
    if len(lines) >= 3 and lines[2].lstrip().startswith('"' + old):
        lines[2] = lines[2].replace('"' + old, '"' + new, 1)
    return "\n".join(lines)


_MARKER = re.compile(r"^#lzgpu-unit (\d+) ([01])\n", re.M)
_AXT_HEAD = re.compile(r"^\d+( \S+ \d+ \d+ \S+ \d+ \d+ [+-])", re.M)


def output_format(args):
    """the --format the command line asks for ('lav' when it names none), lower case"""
    fmt = "lav"
    for a in args:
        a = a.lower()
        if a.startswith("--format=") or a.startswith("--output-format="):
            fmt = a.split("=", 1)[1]
        elif re.match(r"--(lav|maf|axt|waxt|cigar|sam|softsam|rdotplot|text|differences|general|mapping|gfa|blastn|paf)", a):
            fmt = a[2:].replace("=", ":", 1)
    return fmt


def line_oriented(fmt):
    """formats whose output is a header followed by self-contained records, nothing at the end and no state carried from
    one unit to the next beyond AXT's running number (src/maf.c, src/axt.c, src/genpaf.c, src/cigar.c).  SAM is one of
    them as well: its @HD line opens the job and the @SQ lines of the target are printed once, when the first strand
    starts (src/sam.c:195-250, `headerPrinted`) -- ahead of a rank's first unit marker, i.e. in its header."""
    base = re.split(r"[:+-]", fmt.lstrip("~"), 1)[0]
    return base in ("maf", "axt", "waxt", "general", "mapping", "cigar", "differences", "sam", "softsam")


def split_marked(text):
    """output written with LZGPU_UNIT_MARKERS=1 -> (header, [((contig, strand), records)])"""
    ms = list(_MARKER.finditer(text))
    if not ms:
        return text, []
    units = []
    for i, m in enumerate(ms):
        e = ms[i + 1].start() if i + 1 < len(ms) else len(text)
        key = (int(m.group(1)), int(m.group(2)))
        if units and units[-1][0] == key:                          # the same unit marked twice in a row: one block
            units[-1] = (key, units[-1][1] + text[m.end():e])
        else:
            units.append((key, text[m.end():e]))
    return text[:ms[0].start()], units


def merge_marked(texts, fmt="maf", rename=None):
    """ranks' outputs in a line-oriented format -> what one process prints: rank 0's header, then every unit's
    records in file order (queries in file order, + strand before -).  AXT's running alignment number
    (src/axt.c:271-288) is counted again over the merged list."""
    head, allu = None, []
    for r, t in enumerate(texts):
        h, u = split_marked(t)
        if rename and rename[r][0] != rename[r][1]:
            h = h.replace(rename[r][0], rename[r][1])             # the command line in the header comments
        head = h if head is None else head
        allu += u
    keys = [k for k, _ in allu]
    assert len(set(keys)) == len(keys), "a unit was produced by two ranks"
    allu.sort(key=lambda ku: ku[0])
    if fmt.lstrip("~").startswith(("sam", "softsam")):
        # the target's @SQ lines are printed once per process, when its first strand starts (src/sam.c:213-250) -- behind
        # the rank's first unit marker: every rank's first block opens with them; the job has them once, behind @HD / @RG
        sq = []
        for i, (k, u) in enumerate(allu):
            lines = u.splitlines(True)
            mine = [l for l in lines if l.startswith("@SQ\t")]
            if mine:
                sq = sq or mine
                allu[i] = (k, "".join(l for l in lines if not l.startswith("@SQ\t")))
        head = (head or "") + "".join(sq)
    body = "".join(u for _, u in allu)
    if fmt.lstrip("~w").startswith("axt"):
        n = [-1]

        def renumber(m):
            n[0] += 1
            return "%d%s" % (n[0], m.group(1))
        body = _AXT_HEAD.sub(renumber, body)
    return (head or "") + body


def check_supported(target, args, query=None):
    """what the merger cannot put back together is refused before any rank starts"""
    fmt = output_format(args)
    if fmt != "lav" and not line_oriented(fmt):
        raise ValueError("lastz_amd.multi merges LAV, MAF, AXT, SAM, general, cigar and differences output (got %s); run "
                         "the formats the launcher does not know through a single lastz_gpu process" % fmt)
    for a in args:
        if a.startswith("--output=") or a == "--markend":
            raise ValueError("lastz_amd.multi collects the ranks' standard output: %s is not supported" % a)
        # searches that do not go through the seed_hit_search hook carry no unit plan (every rank would compute
        # everything): anchors from a file, chore lists, quantum queries
        if a.startswith("--segments=") or a.startswith("--chores=") or a == "--anyornone" or a.startswith("--anyornone="):
            raise ValueError("lastz_amd.multi shards (query sequence, strand) units of the seed search: %s names work "
                             "that does not come as such units; run it through a single lastz_gpu process" % a)
    # quantum DNA is marked on the sequence specifier (a [quantum] / [quantum=...] action, or a .qdna file), not by an option:
    # a query that merely has "quantum" in its file name is an ordinary query
    for spec in (target, query):
        if spec is None:
            continue
        spath, sact = split_spec(spec)
        actions = [x.strip() for grp in re.findall(r"\[([^\]]*)\]", sact) for x in grp.split(",")]
        if spath.lower().endswith(".qdna") or any(x == "quantum" or x.startswith("quantum=") for x in actions):
            raise ValueError("lastz_amd.multi shards (query sequence, strand) units of the seed search: the quantum sequence %s "
                             "does not go through it; run it through a single lastz_gpu process" % spec)
    tpath, tact = split_spec(target)
    if "[multi]" not in tact and os.path.exists(tpath):
        with open(tpath, "rb") as f:
            magic = f.read(1)
        if magic == b">" and len(fasta_index(tpath)) > 1 and "[" not in tact:
            raise ValueError("the target has several sequences: one lastz_amd.multi run handles one target sequence "
                             "(or a [multi] target); give the sequence as %s[<name>] or use [multi]" % tpath)


def run(target, query, args=(), ranks=2, lastz=DEFAULT_LASTZ, devices=None, transport=None, env=None, keep=False, split=True,
        whole_sequences=None):
    """-> (merged LAV text, [stderr of each rank], plan).  After the call run.last holds {"rank_seconds": [...],
    "owned_bases": [...], "split": bool} of this run."""
    check_supported(target, list(args), query)
    fmt = output_format(list(args))
    qpath, qact = split_spec(query)
    index = fasta_index(qpath)
    lengths = [r[3] for r in index]
    plan = plan_units(lengths, ranks, whole_sequences)
    with open(qpath, "rb") as f:
        is_fasta = f.read(1) == b">"
    split = split and is_fasta and all(a in ("[unmask]", "[nameparse=darkspace]", "[nameparse=full]") for a in re.findall(r"\[[^\]]*\]", qact))
    tmp = tempfile.mkdtemp(prefix="lzgpu_multi_")
    try:
        planf = os.path.join(tmp, "plan.txt")
        with open(planf, "w") as f:
            for r, units in enumerate(plan):
                for (qi, strand) in units:
                    f.write("%d %d %d\n" % (qi + 1, strand, r))
        qfiles = [qpath] * ranks
        if split:
            qfiles = [os.path.join(tmp, "query.rank%d.fa" % r) for r in range(ranks)]
            with ThreadPoolExecutor(max_workers=min(ranks, 8)) as ex:
                list(ex.map(lambda r: write_rank_query(qpath, index, {qi for qi, _ in plan[r]}, qfiles[r]), range(ranks)))
        share = os.path.join(tmp, "share"); os.makedirs(share)       # rendezvous of the table hand-over (lz_share.hip)
        nonce = "%016x" % int.from_bytes(os.urandom(8), "little")
        procs, t0 = [], []
        for r in range(ranks):
            e = dict(os.environ)
            e.update(env or {})
            if fmt != "lav":
                e["LZGPU_UNIT_MARKERS"] = "1"
            e.update({"LZGPU_RANK": str(r), "LZGPU_WORLD": str(ranks), "LZGPU_SHARE_DIR": share, "LZGPU_UNIT_PLAN": planf,
                      "LZGPU_SHARE_NONCE": nonce, "LOCAL_RANK": str(devices[r] if devices else r)})
            if transport:
                e["LZGPU_SHARE_TRANSPORT"] = transport
            t0.append(time.time())
            procs.append(subprocess.Popen([lastz, target, qfiles[r] + qact] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e))
        outs, errs, secs = [None] * ranks, [None] * ranks, [0.0] * ranks

        def wait(r):
            o, e_ = procs[r].communicate()
            secs[r] = time.time() - t0[r]
            outs[r], errs[r] = o.decode(), e_.decode()
            if procs[r].returncode != 0:                        # a rank died: the others must not wait for its table / id
                open(os.path.join(share, "abort"), "wb").close()
                for q in procs:
                    if q.poll() is None:
                        q.kill()
        with ThreadPoolExecutor(max_workers=ranks) as ex:
            list(ex.map(wait, range(ranks)))
        bad = [r for r, p in enumerate(procs) if p.returncode != 0 and p.returncode != -9] or [r for r, p in enumerate(procs) if p.returncode != 0]
        if bad:
            raise RuntimeError("rank %d failed (rc %d): %s" % (bad[0], procs[bad[0]].returncode, errs[bad[0]][-2000:]))
        # start-up self-check (the shim prints the device every rank bound itself to, and stops the run if it is not the one it was given)
        bound = []
        for r in range(ranks):
            m = re.search(r"\[lzgpu\] rank %d of %d: device (\d+)" % (r, ranks), errs[r])
            bound.append(int(m.group(1)) if m else None)
        want = [devices[r] if devices else r for r in range(ranks)]
        if ranks > 1 and any(b is not None and b != w for b, w in zip(bound, want)):
            raise RuntimeError("device binding self-check failed: ranks bound to devices %r, the launcher assigned %r" % (bound, want))
        run.last = {"devices_bound": bound, "rank_seconds": secs, "owned_bases": [sum(lengths[i] for i in {qi for qi, _ in p}) for p in plan], "split": split}
        rename = [(qfiles[r], qpath) for r in range(ranks)]
        merged = merge_lav(outs, rename) if fmt == "lav" else merge_marked(outs, fmt, rename)
        return merged, errs, plan
    finally:
        if not keep:
            shutil.rmtree(tmp, ignore_errors=True)


run.last = {}


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--lastz", default=DEFAULT_LASTZ)
    ap.add_argument("--transport", default=None, help="file: ranks share one device (tests)")
    ap.add_argument("--no-split", action="store_true", help="every rank reads the whole query file (round 2's behaviour)")
    ap.add_argument("rest", nargs=argparse.REMAINDER, help="-- <target> <query> [lastz options]")
    a = ap.parse_args()
    rest = [x for x in a.rest if x != "--"]
    if len(rest) < 2:
        ap.error("need <target> <query>")
    try:
        merged, errs, _ = run(rest[0], rest[1], rest[2:], ranks=a.ranks, lastz=a.lastz, transport=a.transport, split=not a.no_split)
    except ValueError as e:
        sys.exit("lastz_amd.multi: %s" % e)
    sys.stdout.write(merged)
    for r, e in enumerate(errs):
        e = "".join(l for l in e.splitlines(True) if "contains an empty sequence" not in l and not l.startswith(">"))
        if e.strip():
            sys.stderr.write("[rank %d] %s" % (r, e))


if __name__ == "__main__":
    main()
