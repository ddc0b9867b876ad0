"""Shared helpers for the parity tests (test infrastructure)."""
import json
import os
import subprocess
import numpy as np

from oracle import lzo
from lastz_amd import seqio
from lavparse import parse_lav

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
DEFAULT_SEED = "1110100110010101111"


def scoring():
    return lzo.hoxd70_scoring()


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return z["target"], z["query"]


def load_stats(name):
    return json.load(open(os.path.join(GOLDEN, name + ".stats.json")))


def read_hsp_tsv(path):
    """-> list of (name2, start1, end1, start2, end2, strand, score) as the reference printed them"""
    out = []
    for line in open(path):
        f = line.rstrip("\n").split("\t")
        if len(f) < 7:
            continue
        out.append((f[0].strip(), int(f[1]), int(f[2]), int(f[3]), int(f[4]), f[5], int(f[6])))
    return out


def hsps_as_tsv_rows(name2, strand, hsps):
    """our reporter-convention HSPs -> the reference's general-format rows (1-based start, incl. end)"""
    return [(name2, int(h["pos1"] - h["length"]) + 1, int(h["pos1"]), int(h["pos2"] - h["length"]) + 1,
             int(h["pos2"]), strand, int(h["score"])) for h in hsps]


def strands(q):
    return (("+", 0, q), ("-", 1, seqio.revcomp(q)))


def align_segments(al, ops):
    """alignment + editops -> list of gap-free (b1,b2,e1,e2) pieces, i.e. the LAV 'l' lines"""
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
    return [{"score": int(a["s"]), "b": (int(a["beg1"]), int(a["beg2"])), "e": (int(a["end1"]), int(a["end2"])),
             "l": align_segments(a, ops)} for a in aligns]


def lav_blocks(path):
    """golden LAV -> [(contig2, rev2, [blocks without pct])] in file order"""
    out = []
    for st in parse_lav(open(path).read()):
        out.append((st["contig2"], st["rev2"],
                    [{"score": b["score"], "b": b["b"], "e": b["e"], "l": [x[:4] for x in b["l"]]} for b in st["blocks"]]))
    return out


def oracle_hsps(t, q, masked, pattern=DEFAULT_SEED, with_trans=1, table=None, **kw):
    sd = lzo.seed(pattern, with_trans)
    tab = table or lzo.Table(t, sd)
    return lzo.seed_hit_search(tab, q, masked, **kw)


def ref_run(args, **kw):
    binp = lzo.ref_binary()
    return subprocess.check_output([binp] + args, stderr=subprocess.DEVNULL, **kw).decode()


def many_class_scoring():
    """A matrix with more than 8 distinct rows and columns (IUPAC-style partial credit for R/Y/N and a
    separate lower-case penalty): takes the kernels off the 8x8 whole-block scan onto the general path."""
    import numpy as np
    sub, _ = scoring()
    m = sub.copy()
    for amb, members, sc in ((b"R", b"AG", 40), (b"Y", b"CT", 35), (b"K", b"GT", 20), (b"M", b"AC", 15), (b"N", b"ACGT", -10)):
        a = amb[0]
        for b in members:
            m[a, b] = sc; m[b, a] = sc - 5
        m[a, a] = 50 - a % 7
    for b in b"acgt":                                           # lower case: half credit against itself / upper case
        m[b, :] = m[b - 32, :] // 2 - 3; m[:, b] = m[:, b - 32] // 2 - (b % 5)
    return m


def build_emul(tag="", flags=()):
    """tests/emul/libemul.so: the device decomposition (lz_common.hpp, lz_dp_dev.hpp and the host
    orchestration) compiled for the CPU with plain loops in place of the wave primitives.
    tag / flags: a variant build (libemul_<tag>.so with extra compiler flags)."""
    import subprocess
    emul = os.path.join(ROOT, "tests", "emul"); cs = os.path.join(ROOT, "lastz_amd", "csrc")
    so = os.path.join(emul, "libemul%s.so" % ("_" + tag if tag else ""))
    srcs = [os.path.join(emul, "emul_seed.cpp"), os.path.join(emul, "emul_gapped.cpp"), os.path.join(emul, "emul_bounds_plain.cpp"),
            os.path.join(cs, "lz_host.cpp"), os.path.join(cs, "lz_gapped_host.cpp"), os.path.join(cs, "lz_dp_pieces.cpp")]
    deps = srcs + [os.path.join(cs, f) for f in os.listdir(cs) if f.endswith((".hpp", ".h"))] + [os.path.join(ROOT, "include", "lzgpu.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread"] + list(flags) + ["-o", so] + srcs)
    return so
