"""Tiny LAV reader for parity tests (format: /root/reference/lav_format.html).

Returns, per (query index, strand) "s{}" stanza in file order, the list of alignment blocks
  {"score": s, "b": (b1,b2), "e": (e1,e2), "l": [(b1,b2,e1,e2,pct), ...]}
Coordinates are kept exactly as written (1-based, inclusive; minus strand in
reverse-complement coordinates).
"""
import re


def parse_lav(text):
    stanzas = []          # list of dict(name2, strand, blocks)
    cur = None
    lines = text.split("\n")
    i = 0
    while i < len(lines):
        ln = lines[i].strip()
        if ln == "s {":
            l1 = lines[i + 1].strip().split()
            l2 = lines[i + 2].strip().split()
            # "file" start end rev contig
            cur = {"rev2": int(l2[-2]), "contig2": int(l2[-1]), "len1": int(l1[-3]), "len2": int(l2[-3]),
                   "blocks": []}
            stanzas.append(cur)
            i += 3
        elif ln == "a {":
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


def normalize_lav(text):
    """what tools/lav_compare.py treats as equal: drop the first line of the d-stanza (command
    line), strip s/h lines.  Everything else must match byte for byte."""
    out, lines, i = [], text.split("\n"), 0
    while i < len(lines):
        ln = lines[i]
        if ln.strip() == "d {":
            out.append(ln)
            i += 2            # skip the command-line line
            continue
        out.append(ln)
        i += 1
    return "\n".join(out)
