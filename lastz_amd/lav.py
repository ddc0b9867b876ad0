"""LAV text <-> alignment blocks, for checks that compare what the library returns with what the bound CLI prints
(bench.py, tests).  Format: lav_format.html of the reference; coordinates 1-based inclusive, the minus strand in
reverse-complement coordinates -- the convention lzgpu_gapped_extend's lz_align records already use."""


def parse(text):
    """-> [{"rev2", "contig2", "len1", "len2", "blocks": [{"score", "b", "e", "l": [(b1, b2, e1, e2, pct)]}]}] per s-stanza, file order"""
    stanzas, cur = [], None
    lines = text.split("\n")
    i = 0
    while i < len(lines):
        ln = lines[i].strip()
        if ln == "s {":
            l1 = lines[i + 1].strip().split(); l2 = lines[i + 2].strip().split()
            cur = {"rev2": int(l2[-2]), "contig2": int(l2[-1]), "len1": int(l1[-3]), "len2": int(l2[-3]), "blocks": []}
            stanzas.append(cur)
            i += 3
            continue
        if ln == "a {":
            blk = {"l": []}
            i += 1
            while lines[i].strip() != "}":
                f = lines[i].split()
                if f[0] == "s":
                    blk["score"] = int(f[1])
                elif f[0] == "b":
                    blk["b"] = (int(f[1]), int(f[2]))
                elif f[0] == "e":
                    blk["e"] = (int(f[1]), int(f[2]))
                elif f[0] == "l":
                    blk["l"].append(tuple(int(x) for x in f[1:6]))
                i += 1
            cur["blocks"].append(blk)
        i += 1
    return stanzas


def pieces(al, ops):
    """one lz_align record + the edit ops -> its gap-free pieces (b1, b2, e1, e2): the 'l' lines of its LAV block"""
    res = []
    p1, p2 = int(al["beg1"]), int(al["beg2"])
    for k in range(int(al["script_off"]), int(al["script_off"]) + int(al["script_len"])):
        op, rpt = int(ops[k]) & 3, int(ops[k]) >> 2
        if op == 3:
            res.append((p1, p2, p1 + rpt - 1, p2 + rpt - 1)); p1 += rpt; p2 += rpt
        elif op == 1:
            p2 += rpt
        else:
            p1 += rpt
    return res


def blocks_of(aligns, ops):
    return [{"score": int(a["s"]), "b": (int(a["beg1"]), int(a["beg2"])), "e": (int(a["end1"]), int(a["end2"])), "l": pieces(a, ops)} for a in aligns]


def compare(per_strand, lav_text):
    """per_strand: [(aligns, ops) of the + strand, of the - strand] as lzgpu_gapped_extend(_batch) returns them; lav_text: the LAV of
    the same pair.  Field by field: score, begin, end and every gap-free piece of every block, in the order printed.
    -> {"equal": bool, "blocks": n, "pieces": n, "first_difference": None | str}"""
    st = {s["rev2"]: s["blocks"] for s in parse(lav_text)}
    n_blocks = n_pieces = 0
    for rev, (al, ops) in enumerate(per_strand):
        mine = blocks_of(al, ops)
        theirs = st.get(rev, [])
        if len(mine) != len(theirs):
            return {"equal": False, "blocks": n_blocks, "pieces": n_pieces, "first_difference": "strand %d: %d alignments here, %d blocks in the LAV" % (rev, len(mine), len(theirs))}
        for k, (m, t) in enumerate(zip(mine, theirs)):
            tl = [x[:4] for x in t["l"]]
            if m["score"] != t["score"] or m["b"] != t["b"] or m["e"] != t["e"] or m["l"] != tl:
                return {"equal": False, "blocks": n_blocks, "pieces": n_pieces, "first_difference": "strand %d block %d: %r vs %r" % (rev, k, {**m, "l": m["l"][:3]}, {**t, "l": tl[:3]})}
            n_blocks += 1; n_pieces += len(tl)
    return {"equal": True, "blocks": n_blocks, "pieces": n_pieces, "first_difference": None}


# ---- the other direction: alignments -> the LAV text the reference writes for them (src/lav.c:57-127, 235-300), for the plain case the
# bench pairs are (one target sequence, one query sequence, both strands, a 4 x 4 matrix over A, C, G, T).  With it the alignments that
# lzgpu_gapped_extend_batch RETURNS can be fingerprinted exactly like the file the pristine reference wrote for the pair: every score,
# coordinate, piece and percent-identity column, in order -- not a count (VERDICT r4 #3d).
def _matrix_text(sub):
    """print_score_matrix over "ACGT" (src/dna_utilities.c:1996-2083, integer scores: fields of width 4 behind a blank; the row
    characters are not printed when both alphabets are printable), the layout of the d-stanza"""
    import numpy as np
    sub = np.asarray(sub)
    lines = [" " + "".join(" %4s" % chr(c) for c in b"ACGT")]
    for a in b"ACGT":
        lines.append(" " + "".join(" %4d" % int(sub[a, b]) for b in b"ACGT"))
    return lines


def render(target, query_by_strand, per_strand, sub, name1="t.fa", name2="q.fa", header1=">target", header2=">query",
           gap_open=400, gap_extend=30, hsp_thresh=3000, gapped_thresh=3000, command_line="lastz"):
    """target: uint8 array; query_by_strand: [query, its reverse complement]; per_strand: [(aligns, ops)] in the same order;
    -> the LAV text (its first d-stanza line holds the command line: every fingerprint of a LAV leaves that line out)"""
    import numpy as np
    out = ["#:lav", "d {", '  "%s' % command_line] + _matrix_text(sub)
    out.append('  O = %d, E = %d, K = %d, L = %d, M = 0"' % (gap_open, gap_extend, hsp_thresh, gapped_thresh))
    out.append("}")
    tu = np.asarray(target).copy(); tu[(tu >= 97) & (tu <= 122)] -= 32         # dna_toupper: a percent-identity match ignores case
    for rev, (q, (al, ops)) in enumerate(zip(query_by_strand, per_strand)):
        qu = np.asarray(q).copy(); qu[(qu >= 97) & (qu <= 122)] -= 32
        out += ["#:lav", "s {", '  "%s" 1 %d 0 1' % (name1, len(tu)), '  "%s%s" 1 %d %d 1' % (name2, "-" if rev else "", len(qu), rev), "}",
                "h {", '   "%s"' % header1, '   "%s%s"' % (header2, " (reverse complement)" if rev else ""), "}"]
        ops = np.asarray(ops)
        for a in al:
            b1, b2, e1, e2 = int(a["beg1"]), int(a["beg2"]), int(a["end1"]), int(a["end2"])
            out += ["a {", "  s %d" % int(a["s"]), "  b %d %d" % (b1, b2), "  e %d %d" % (e1, e2)]
            k, k_end = int(a["script_off"]), int(a["script_off"]) + int(a["script_len"])
            i = j = 0
            height, width = e1 - b1 + 1, e2 - b2 + 1
            while i < height or j < width:
                run = 0
                while k < k_end and (int(ops[k]) & 3) == 3:          # a run of substitution ops (possibly none: an insert behind a delete)
                    run += int(ops[k]) >> 2; k += 1
                match = int((tu[b1 - 1 + i:b1 - 1 + i + run] == qu[b2 - 1 + j:b2 - 1 + j + run]).sum()) if run else 0
                pct = (200 * match + run) // (2 * run) if run else 0
                out.append("  l %d %d %d %d %d" % (b1 + i, b2 + j, b1 + i + run - 1, b2 + j + run - 1, pct))
                i += run; j += run
                if (i < height or j < width) and k < k_end:          # ONE indel op, then the next run
                    op, rpt = int(ops[k]) & 3, int(ops[k]) >> 2; k += 1
                    if op == 1:
                        j += rpt
                    else:
                        i += rpt
                elif k >= k_end and (i < height or j < width):
                    raise ValueError("edit script shorter than its alignment")
            out.append("}")
    out += ["m {", "  n 0", "}", "#:eof", ""]
    return "\n".join(out)


def fingerprint(text):
    """SHA-256 of a LAV without the command line (line 1 of its d-stanza): what tests/golden/bench*.sha.json pins"""
    import hashlib
    lines = text.split("\n")
    for i, ln in enumerate(lines):
        if ln.startswith("d {"):
            del lines[i + 1]
            break
    return hashlib.sha256("\n".join(lines).encode()).hexdigest()
