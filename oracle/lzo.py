"""ctypes binding of the CPU oracle (oracle/liblz_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product (lastz_amd/) never imports this.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MAX_PARTS, MAX_PROBES = 16, 128
NEG_INF = int(0.9 * (-0x7FFFFFFF - 1))
VERY_BAD = -((NEG_INF - (-0x7FFFFFFF - 1)) // 2)


class Seed(C.Structure):
    _fields_ = [("length", C.c_int), ("weight", C.c_int), ("num_parts", C.c_int),
                ("shift", C.c_int * MAX_PARTS), ("mask", C.c_uint32 * MAX_PARTS),
                ("with_trans", C.c_int), ("num_flips", C.c_int), ("flips", C.c_uint32 * 32),
                ("num_probes", C.c_int), ("probe_xor", C.c_uint32 * MAX_PROBES)]


class PosTable(C.Structure):
    _fields_ = [("last", C.POINTER(C.c_uint32)), ("prev", C.POINTER(C.c_uint32)),
                ("word_entries", C.c_uint32), ("prev_entries", C.c_uint32),
                ("start", C.c_uint32), ("end", C.c_uint32), ("adj_start", C.c_uint32),
                ("step", C.c_uint32), ("words_in_table", C.c_uint64)]


class SearchStats(C.Structure):
    _fields_ = [("words", C.c_uint64), ("raw_hits", C.c_uint64), ("extensions", C.c_uint64),
                ("bp_extended", C.c_uint64), ("hsps", C.c_uint64)]


class GappedStats(C.Structure):
    _fields_ = [("anchors", C.c_uint64), ("anchors_extended", C.c_uint64),
                ("extensions", C.c_uint64), ("dp_cells", C.c_uint64),
                ("max_rows", C.c_uint32), ("max_cols", C.c_uint32), ("truncations", C.c_uint64)]


HSP_DTYPE = np.dtype([("pos1", "<u4"), ("pos2", "<u4"), ("length", "<u4"), ("score", "<i4")])
SEG_DTYPE = np.dtype([("pos1", "<u4"), ("pos2", "<u4"), ("length", "<u4"), ("s", "<i4"), ("id", "<i4")])
ALIGN_DTYPE = np.dtype([("beg1", "<u4"), ("beg2", "<u4"), ("end1", "<u4"), ("end2", "<u4"),
                        ("s", "<i4"), ("script_len", "<u4"), ("script_off", "<u4")])


def build():
    """(Re)build liblz_oracle.so (and oracle/_ref when /root/reference is present)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "liblz_oracle.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    L = C.CDLL(path)
    L.lzo_seed_from_pattern.argtypes = [C.c_char_p, C.c_int, C.POINTER(Seed)]
    L.lzo_apply_seed.argtypes = [C.POINTER(Seed), C.c_uint64]
    L.lzo_apply_seed.restype = C.c_uint32
    L.lzo_entropy.restype = C.c_double
    L.lzo_entropy.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.lzo_build_position_table.restype = C.POINTER(PosTable)
    L.lzo_build_position_table.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                           C.c_void_p, C.POINTER(Seed), C.c_uint32]
    L.lzo_free_position_table.argtypes = [C.POINTER(PosTable)]
    L.lzo_position_table_to_csr.restype = C.c_uint64
    L.lzo_position_table_to_csr.argtypes = [C.POINTER(PosTable), C.c_void_p, C.c_void_p]
    L.lzo_seed_hit_search.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(PosTable),
                                      C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.c_void_p, C.POINTER(Seed), C.c_void_p, C.c_int32,
                                      C.c_int32, C.c_int, C.c_int, C.c_uint32,
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                      C.POINTER(SearchStats)]
    L.lzo_free.argtypes = [C.c_void_p]
    L.lzo_reduce_to_points.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    L.lzo_gapped_extend.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                    C.c_int32, C.c_int32, C.c_void_p, C.c_uint32, C.c_int32,
                                    C.c_int, C.c_int32, C.c_uint32,
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                    C.POINTER(GappedStats)]
    L.lzo_gapped_extend_opts.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                         C.c_int32, C.c_int32, C.c_void_p, C.c_uint32, C.c_int32,
                                         C.c_int, C.c_int, C.c_int32, C.c_uint32,
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                         C.POINTER(GappedStats)]
    _LIB = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def seed(pattern="1110100110010101111", with_trans=1):
    s = Seed()
    rc = lib().lzo_seed_from_pattern(pattern.encode(), with_trans, C.byref(s))
    if rc != 0:
        raise ValueError(f"bad seed pattern {pattern!r}: {rc}")
    return s


def hoxd70_scoring(bad=-1000, fill=-100):
    """(scoring, maskedScoring) as int32[256,256] -- lastz defaults."""
    t = (C.c_int32 * 16)()
    lib().lzo_hoxd70(t)
    sub = np.zeros((256, 256), dtype=np.int32)
    lib().lzo_dna_score_set(t, bad, fill, _ptr(sub))
    masked = np.zeros((256, 256), dtype=np.int32)
    lib().lzo_masked_score_set(_ptr(sub), _ptr(masked))
    return sub, masked


def upper_nuc_to_bits():
    t = np.zeros(256, dtype=np.int8)
    lib().lzo_upper_nuc_to_bits(_ptr(t))
    return t


def nul_terminated(seq):
    """uint8 array with one trailing NUL (the reference's seq.v layout)."""
    a = np.frombuffer(bytes(seq), dtype=np.uint8) if not isinstance(seq, np.ndarray) else seq
    out = np.zeros(len(a) + 1, dtype=np.uint8)
    out[:len(a)] = a
    return out


class Table:
    def __init__(self, t, sd, step=1, start=0, end=0, ctb=None):
        self.t = nul_terminated(t)
        self.tlen = len(self.t) - 1
        self.sd = sd
        self.ctb = upper_nuc_to_bits() if ctb is None else ctb
        self.pt = lib().lzo_build_position_table(_ptr(self.t), self.tlen, start, end,
                                                 _ptr(self.ctb), C.byref(sd), step)
        if not self.pt:
            raise ValueError("bad table interval")

    def csr(self):
        n = int(self.pt.contents.words_in_table)
        ws = np.zeros(int(self.pt.contents.word_entries) + 1, dtype=np.uint32)
        wp = np.zeros(max(n, 1), dtype=np.uint32)
        got = lib().lzo_position_table_to_csr(self.pt, _ptr(ws), _ptr(wp))
        assert got == n
        return ws, wp[:n]

    def __del__(self):
        try:
            lib().lzo_free_position_table(self.pt)
        except Exception:
            pass


def seed_hit_search(table, q, masked_sub, xdrop=910, hsp_threshold=3000, entropic=True,
                    mode=0, start=0, end=0, diag_hash_size=65536):
    qa = nul_terminated(q)
    qlen = len(qa) - 1
    out = C.c_void_p()
    n = C.c_uint64()
    st = SearchStats()
    rc = lib().lzo_seed_hit_search(_ptr(table.t), table.tlen, table.pt, _ptr(qa), qlen, start, end,
                                   _ptr(table.ctb), C.byref(table.sd), _ptr(masked_sub),
                                   xdrop, hsp_threshold, int(entropic), mode, diag_hash_size,
                                   C.byref(out), C.byref(n), C.byref(st))
    if rc != 0:
        raise RuntimeError(f"lzo_seed_hit_search rc={rc}")
    res = np.zeros(n.value, dtype=HSP_DTYPE)
    if n.value:
        C.memmove(_ptr(res), out, n.value * HSP_DTYPE.itemsize)
    lib().lzo_free(out)
    stats = {k: int(getattr(st, k)) for k, _ in SearchStats._fields_}
    return res, stats


def hsps_to_segments(hsps, seg_id=0):
    segs = np.zeros(len(hsps), dtype=SEG_DTYPE)
    segs["pos1"] = hsps["pos1"] - hsps["length"]
    segs["pos2"] = hsps["pos2"] - hsps["length"]
    segs["length"] = hsps["length"]
    segs["s"] = hsps["score"]
    segs["id"] = seg_id
    return segs


def reduce_to_points(t, q, sub, segs):
    ta, qa = nul_terminated(t), nul_terminated(q)
    segs = segs.copy()
    lib().lzo_reduce_to_points(_ptr(ta), _ptr(qa), _ptr(sub), _ptr(segs), len(segs))
    return segs


def gapped_extend(t, q, sub, anchors, gap_open=400, gap_extend=30, ydrop=9400, trim_to_peak=True,
                  score_thresh=3000, tb_size=0, all_bounds=False):
    """anchors: SEG_DTYPE array already reduced to points.  Returns (aligns, ops, stats)."""
    ta, qa = nul_terminated(t), nul_terminated(q)
    anchors = anchors.copy()
    out = C.c_void_p(); n = C.c_uint64(); ops = C.c_void_p(); nops = C.c_uint64()
    st = GappedStats()
    rc = lib().lzo_gapped_extend_opts(_ptr(ta), len(ta) - 1, _ptr(qa), len(qa) - 1, _ptr(sub),
                                 gap_open, gap_extend, _ptr(anchors), len(anchors), ydrop,
                                 int(trim_to_peak), int(all_bounds), score_thresh, tb_size,
                                 C.byref(out), C.byref(n), C.byref(ops), C.byref(nops), C.byref(st))
    if rc != 0:
        raise RuntimeError(f"lzo_gapped_extend rc={rc}")
    al = np.zeros(n.value, dtype=ALIGN_DTYPE)
    op = np.zeros(nops.value, dtype=np.uint32)
    if n.value:
        C.memmove(_ptr(al), out, n.value * ALIGN_DTYPE.itemsize)
    if nops.value:
        C.memmove(_ptr(op), ops, nops.value * 4)
    lib().lzo_free(out); lib().lzo_free(ops)
    stats = {k: int(getattr(st, k)) for k, _ in GappedStats._fields_}
    return al, op, stats


# ---- the pristine reference binary (oracle/_ref), when present -------------------------------

def ref_binary(stats=False):
    p = os.path.join(_HERE, "_ref", "lastz_stats" if stats else "lastz")
    return p if os.path.exists(p) else None
