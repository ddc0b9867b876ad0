"""ctypes binding of liblzgpu.so (include/lzgpu.h) for tests and bench.py.

Plumbing only: the product is the C-ABI library.  There is no CPU fallback here -- if the
library is missing or no gfx950 device is usable, calls raise.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LZGPU_LIB") or os.path.join(_HERE, "liblzgpu.so")   # LZGPU_LIB: try a variant build

MAX_PARTS, MAX_PROBES = 16, 128

NH_REASONS = {1: "SEED", 2: "SCORE_CLASSES", 3: "HITS_OVERFLOW", 4: "HSP_OVERFLOW", 5: "SIZE",
              6: "IDENTICAL", 7: "UNSUPPORTED", 8: "PAIRED_LIMIT"}


class SeedDesc(C.Structure):
    _fields_ = [("length", C.c_int32), ("weight_bits", C.c_int32), ("num_parts", C.c_int32),
                ("shift", C.c_int32 * MAX_PARTS), ("mask", C.c_uint32 * MAX_PARTS),
                ("num_probes", C.c_int32), ("probe_xor", C.c_uint32 * MAX_PROBES)]


class TableGeom(C.Structure):
    _fields_ = [("tlen", C.c_uint32), ("start", C.c_uint32), ("end", C.c_uint32), ("step", C.c_uint32),
                ("num_words", C.c_uint64), ("seed", SeedDesc), ("char_to_bits", C.c_int8 * 256)]


class SearchArgs(C.Structure):
    _fields_ = [("query", C.c_void_p), ("qlen", C.c_uint32), ("query_slot", C.c_int32),
                ("start", C.c_uint32), ("end", C.c_uint32), ("sub", C.c_void_p),
                ("xdrop", C.c_int32), ("hsp_threshold", C.c_int32), ("entropic", C.c_int32),
                ("extend", C.c_int32)]


class GappedArgs(C.Structure):
    _fields_ = [("query", C.c_void_p), ("qlen", C.c_uint32), ("query_slot", C.c_int32),
                ("sub", C.c_void_p), ("gap_open", C.c_int32), ("gap_extend", C.c_int32),
                ("ydrop", C.c_int32), ("score_thresh", C.c_int32), ("traceback_bytes", C.c_uint32),
                ("anchors", C.c_void_p), ("n_anchors", C.c_uint32), ("reduce", C.c_int32),
                ("sep1", C.c_void_p), ("n_sep1", C.c_uint32), ("sep2", C.c_void_p), ("n_sep2", C.c_uint32),
                ("strands_differ", C.c_int32), ("inhibit_trivial", C.c_int32),
                ("t_off", C.c_uint32), ("t_len", C.c_uint32), ("q_off", C.c_uint32), ("q_len", C.c_uint32),
                ("all_bounds", C.c_int32), ("no_trim", C.c_int32), ("max_paired_bases", C.c_uint64)]


class WindowSearchArgs(C.Structure):
    _fields_ = [("query", C.c_void_p), ("qlen", C.c_uint32), ("query_slot", C.c_int32), ("sub", C.c_void_p),
                ("xdrop", C.c_int32), ("hsp_threshold", C.c_int32), ("seed", C.c_void_p), ("char_to_bits", C.c_void_p),
                ("windows", C.c_void_p), ("n_windows", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [("words", C.c_uint64), ("raw_hits", C.c_uint64), ("extensions", C.c_uint64),
                ("bp_extended", C.c_uint64), ("hsps", C.c_uint64), ("dp_cells", C.c_uint64),
                ("gapped_extensions", C.c_uint64), ("anchors_extended", C.c_uint64), ("truncated_extensions", C.c_uint64),
                ("dp_rows", C.c_uint64)]


HSP_DTYPE = np.dtype([("pos1", "<u4"), ("pos2", "<u4"), ("length", "<u4"), ("score", "<i4")])
SEG_DTYPE = np.dtype([("pos1", "<u4"), ("pos2", "<u4"), ("length", "<u4"), ("s", "<i4"), ("id", "<i4")])
ALIGN_DTYPE = np.dtype([("beg1", "<u4"), ("beg2", "<u4"), ("end1", "<u4"), ("end2", "<u4"),
                        ("s", "<i4"), ("script_len", "<u4"), ("script_off", "<u4")])

# every symbol include/lzgpu.h declares (tests check that the library exports all of them)
EXPORTS = ["lzgpu_seed_from_pattern", "lzgpu_probe", "lzgpu_init", "lzgpu_device_index", "lzgpu_init_async", "lzgpu_shutdown", "lzgpu_free",
           "lzgpu_last_error", "lzgpu_table_prepare", "lzgpu_table_export", "lzgpu_table_rebuild", "lzgpu_table_num_words",
           "lzgpu_table_geom", "lzgpu_table_adopt", "lzgpu_table_buffers", "lzgpu_table_commit", "lzgpu_table_share", "lzgpu_table_save", "lzgpu_table_load", "lzgpu_device_copy",
           "lzgpu_seed_hit_search", "lzgpu_query_upload", "lzgpu_target_upload", "lzgpu_gapped_extend", "lzgpu_gapped_extend_batch", "lzgpu_window_search",
           "lzgpu_counters_reset", "lzgpu_counters_get", "lzgpu_profile_enable", "lzgpu_profile_reset",
           "lzgpu_profile_get", "lzgpu_set_hit_capacity", "lzgpu_set_hsp_capacity", "lzgpu_set_dp_slot", "lzgpu_set_dp_window", "lzgpu_dp_longest",
           "lzgpu_set_bucket_owner", "lzgpu_last_hsp_order", "lzgpu_last_scan_mode", "lzgpu_set_scan_mode", "lzgpu_reduce_to_chain", "lzgpu_reduce_to_chain_batch"]


class LzGpuError(RuntimeError):
    pass


class NotHandled(Exception):
    """rc > 0: the library declined; the caller would run the reference CPU routine."""
    def __init__(self, rc):
        super().__init__(f"not handled: {NH_REASONS.get(rc, rc)}")
        self.rc = rc


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Lib:
    """One process <-> one GPU.  `prefix`/`path` let tests bind the CPU emulation harness
    (tests/emul) through the same wrapper; the product always uses the defaults."""

    def __init__(self, path=LIB_PATH, prefix="lzgpu_"):
        if not os.path.exists(path):
            raise LzGpuError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        self.L = C.CDLL(path)
        self.px = prefix
        f = self._f
        f("seed_hit_search").argtypes = [C.POINTER(SearchArgs), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        f("table_prepare").argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                       C.POINTER(SeedDesc), C.c_uint32]
        f("set_hit_capacity").argtypes = [C.c_uint64]
        f("counters_get").argtypes = [C.POINTER(Counters)]
        f("free").argtypes = [C.c_void_p]
        if prefix == "lzgpu_":
            self.L.lzgpu_seed_from_pattern.argtypes = [C.c_char_p, C.c_int, C.POINTER(SeedDesc)]
            self.L.lzgpu_last_error.restype = C.c_char_p
            self.L.lzgpu_table_num_words.restype = C.c_uint64
            self.L.lzgpu_table_export.argtypes = [C.c_void_p, C.c_void_p]
            self.L.lzgpu_table_geom.argtypes = [C.POINTER(TableGeom)]
            self.L.lzgpu_table_adopt.argtypes = [C.POINTER(TableGeom)]
            self.L.lzgpu_table_buffers.argtypes = [C.POINTER(C.c_void_p * 3), C.POINTER(C.c_uint64 * 3)]
            self.L.lzgpu_device_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
            self.L.lzgpu_query_upload.argtypes = [C.c_int32, C.c_void_p, C.c_uint32]
            self.L.lzgpu_target_upload.argtypes = [C.c_void_p, C.c_uint32]
            self.L.lzgpu_gapped_extend_batch.argtypes = [C.POINTER(GappedArgs), C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                                                         C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
            self.L.lzgpu_gapped_extend.argtypes = [C.POINTER(GappedArgs), C.POINTER(C.c_void_p),
                                                   C.POINTER(C.c_uint64), C.POINTER(C.c_void_p),
                                                   C.POINTER(C.c_uint64)]
            self.L.lzgpu_set_hsp_capacity.argtypes = [C.c_uint64]
            self.L.lzgpu_profile_get.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64),
                                                 C.POINTER(C.c_double)]
        self._keep = {}

    def _f(self, name):
        return getattr(self.L, self.px + name)

    def _check(self, rc, what):
        if rc > 0:
            raise NotHandled(rc)
        if rc < 0:
            msg = self.L.lzgpu_last_error().decode() if self.px == "lzgpu_" else ""
            raise LzGpuError(f"{what} failed rc={rc}: {msg}")

    # ---- lifecycle
    def init(self, device=-1):
        self._check(self.L.lzgpu_init(device), "lzgpu_init")

    def probe(self):
        return self.L.lzgpu_probe()

    def device_index(self):
        """the device this process's library is bound to (-1 before init)"""
        return int(self.L.lzgpu_device_index())

    def shutdown(self):
        self.L.lzgpu_shutdown()

    # ---- seeds
    def seed(self, pattern="1110100110010101111", with_trans=1):
        sd = SeedDesc()
        fn = self.L.lzgpu_seed_from_pattern      # (the emulation harness links the same lz_host.cpp)
        fn.argtypes = [C.c_char_p, C.c_int, C.POINTER(SeedDesc)]
        rc = fn(pattern.encode(), with_trans, C.byref(sd))
        self._check(rc, "lzgpu_seed_from_pattern")
        return sd

    # ---- B1
    def table_prepare(self, t, sd, char_to_bits, step=1, start=0, end=0):
        t = np.ascontiguousarray(t, dtype=np.uint8)
        ctb = np.ascontiguousarray(char_to_bits, dtype=np.int8)
        self._keep["ctb"] = ctb
        self._check(self._f("table_prepare")(_ptr(t), len(t), start, end, _ptr(ctb), C.byref(sd), step),
                    "lzgpu_table_prepare")
        self._weight = sd.weight_bits

    def table_rebuild(self):
        self._check(self.L.lzgpu_table_rebuild(), "lzgpu_table_rebuild")

    def table_num_words(self):
        return int(self.L.lzgpu_table_num_words())

    def table_export(self, prev_entries):
        last = np.zeros(1 << self._weight, dtype=np.uint32)
        prev = np.zeros(prev_entries, dtype=np.uint32)
        self._check(self.L.lzgpu_table_export(_ptr(last), _ptr(prev)), "lzgpu_table_export")
        return last, prev

    def table_geom(self):
        g = TableGeom()
        self._check(self.L.lzgpu_table_geom(C.byref(g)), "lzgpu_table_geom")
        return g

    def table_adopt(self, g):
        self._check(self.L.lzgpu_table_adopt(C.byref(g)), "lzgpu_table_adopt")
        self._weight = g.seed.weight_bits

    def table_buffers(self):
        p = (C.c_void_p * 3)()
        b = (C.c_uint64 * 3)()
        self._check(self.L.lzgpu_table_buffers(C.byref(p), C.byref(b)), "lzgpu_table_buffers")
        return [(int(p[i] or 0), int(b[i])) for i in range(3)]

    def device_copy(self, dst, src, nbytes):
        self._check(self.L.lzgpu_device_copy(dst, src, nbytes), "lzgpu_device_copy")

    def table_commit(self):
        self._check(self.L.lzgpu_table_commit(), "lzgpu_table_commit")

    # ---- B2
    def query_upload(self, slot, q):
        q = np.ascontiguousarray(q, dtype=np.uint8)
        self._check(self.L.lzgpu_query_upload(slot, _ptr(q), len(q)), "lzgpu_query_upload")
        self._keep[("qlen", slot)] = len(q)

    def seed_hit_search(self, sub, q=None, slot=-1, xdrop=910, hsp_threshold=3000, entropic=True,
                        extend=True, start=0, end=0):
        a = SearchArgs()
        sub = np.ascontiguousarray(sub, dtype=np.int32)
        if q is not None:
            q = np.ascontiguousarray(q, dtype=np.uint8)
            a.query, a.qlen = q.ctypes.data, len(q)
        else:
            a.query, a.qlen = None, self._keep[("qlen", slot)]
        a.query_slot, a.start, a.end = slot, start, end
        a.sub, a.xdrop, a.hsp_threshold = sub.ctypes.data, xdrop, hsp_threshold
        a.entropic, a.extend = int(entropic), int(extend)
        out = C.c_void_p()
        n = C.c_uint64()
        self._check(self._f("seed_hit_search")(C.byref(a), C.byref(out), C.byref(n)), "lzgpu_seed_hit_search")
        res = np.zeros(n.value, dtype=HSP_DTYPE)
        if n.value:
            C.memmove(_ptr(res), out, n.value * HSP_DTYPE.itemsize)
        self._f("free")(out)
        return res

    def set_bucket_owner(self, n_owners, owner):
        self._check(self._f("set_bucket_owner")(C.c_uint32(n_owners), C.c_uint32(owner)), "lzgpu_set_bucket_owner")

    def table_save(self, path):
        self._check(self.L.lzgpu_table_save(str(path).encode()), "lzgpu_table_save")

    def table_load(self, path):
        self._check(self.L.lzgpu_table_load(str(path).encode()), "lzgpu_table_load")

    def last_scan_mode(self):
        return int(self.L.lzgpu_last_scan_mode())

    def set_scan_mode(self, min_mode):
        self._check(self.L.lzgpu_set_scan_mode(C.c_int(min_mode)), "lzgpu_set_scan_mode")

    def last_hsp_order(self, n):
        """(n, 2) uint64 sort words of the HSPs the last search returned (see include/lzgpu.h)"""
        out = np.zeros((n, 2), dtype=np.uint64)
        self._check(self._f("last_hsp_order")(_ptr(out), C.c_uint64(n)), "lzgpu_last_hsp_order")
        return out

    def target_upload(self, t):
        t = np.ascontiguousarray(t, dtype=np.uint8)
        self._check(self.L.lzgpu_target_upload(_ptr(t), len(t)), "lzgpu_target_upload")

    # ---- B3
    def _gapped_args(self, keep, sub, anchors, q=None, slot=-1, gap_open=400, gap_extend=30, ydrop=9400, score_thresh=3000,
                     traceback_bytes=0, reduce=True, sep1=None, sep2=None, strands_differ=False, inhibit_trivial=False,
                     t_off=0, t_len=0, q_off=0, q_len=0, all_bounds=False, no_trim=False, max_paired_bases=0):
        a = GappedArgs()
        a.strands_differ, a.inhibit_trivial = int(strands_differ), int(inhibit_trivial)
        a.all_bounds, a.no_trim, a.max_paired_bases = int(all_bounds), int(no_trim), int(max_paired_bases)
        a.t_off, a.t_len, a.q_off, a.q_len = t_off, t_len, q_off, q_len
        if sep1 is not None:
            sep1 = np.ascontiguousarray(sep1, dtype=np.uint32); a.sep1, a.n_sep1 = sep1.ctypes.data, len(sep1)
        if sep2 is not None:
            sep2 = np.ascontiguousarray(sep2, dtype=np.uint32); a.sep2, a.n_sep2 = sep2.ctypes.data, len(sep2)
        sub = np.ascontiguousarray(sub, dtype=np.int32)
        anchors = np.ascontiguousarray(anchors.copy(), dtype=SEG_DTYPE)
        if q is not None:
            q = np.ascontiguousarray(q, dtype=np.uint8)
            a.query, a.qlen = q.ctypes.data, len(q)
        else:
            a.query, a.qlen = None, self._keep[("qlen", slot)]
        a.query_slot = slot
        a.sub, a.gap_open, a.gap_extend, a.ydrop = sub.ctypes.data, gap_open, gap_extend, ydrop
        a.score_thresh, a.traceback_bytes = score_thresh, traceback_bytes
        a.anchors, a.n_anchors, a.reduce = anchors.ctypes.data, len(anchors), int(reduce)
        keep += [sub, anchors, q, sep1, sep2]                  # alive until the call returns
        return a

    @staticmethod
    def _gapped_out(L, out, n, ops, nops):
        al = np.zeros(n, dtype=ALIGN_DTYPE)
        op = np.zeros(nops, dtype=np.uint32)
        if n:
            C.memmove(_ptr(al), out, n * ALIGN_DTYPE.itemsize)
        if nops:
            C.memmove(_ptr(op), ops, nops * 4)
        L.lzgpu_free(out); L.lzgpu_free(ops)
        return al, op

    def gapped_extend(self, sub, anchors, **kw):
        """sep1 / sep2: positions of the NUL bytes bounding the partitions of a [multi] target / query;
        t_off / t_len / q_off / q_len: a rectangle of the two sequences as the whole problem"""
        keep = []
        a = self._gapped_args(keep, sub, anchors, **kw)
        out = C.c_void_p(); n = C.c_uint64(); ops = C.c_void_p(); nops = C.c_uint64()
        self._check(self.L.lzgpu_gapped_extend(C.byref(a), C.byref(out), C.byref(n), C.byref(ops), C.byref(nops)),
                    "lzgpu_gapped_extend")
        return self._gapped_out(self.L, out, n.value, ops, nops.value)

    def gapped_extend_batch(self, sub, problems):
        """problems: [dict(anchors=..., slot=... | q=..., ...)] sharing `sub` and the gap / y-drop parameters; the DPs of all
        of them share the launches.  -> [(alignments, ops)] in the order of `problems`"""
        keep, n = [], len(problems)
        arr = (GappedArgs * n)()
        for k, pr in enumerate(problems):
            pr = dict(pr)
            arr[k] = self._gapped_args(keep, sub, pr.pop("anchors"), **pr)
        out = (C.c_void_p * n)(); no = (C.c_uint64 * n)(); ops = (C.c_void_p * n)(); nops = (C.c_uint64 * n)()
        self._check(self.L.lzgpu_gapped_extend_batch(arr, n, out, no, ops, nops), "lzgpu_gapped_extend_batch")
        return [self._gapped_out(self.L, out[k], no[k], ops[k], nops[k]) for k in range(n)]

    # ---- N2: chaining (host routine of the library; needs no device)
    def reduce_to_chain(self, anchors, chain_diag=0, chain_anti=0, scale=100, overlap_sub=91, diag_pen=None, anti_pen=None):
        """anchors: SEG_DTYPE array of one (query, strand) -> (indices of the chain's members in the reference's order, chain score);
        src/chain.c:497 with chain_connect_penalty (src/lastz.c:3687); overlap_sub = sub[rowChars[0]][colChars[0]]"""
        anchors = np.ascontiguousarray(anchors, dtype=SEG_DTYPE)
        a = (C.c_int32 * 6)(chain_diag if diag_pen is None else diag_pen, chain_anti if anti_pen is None else anti_pen,
                            chain_diag, chain_anti, scale, overlap_sub)
        kept = C.c_void_p(); n = C.c_uint32(); best = C.c_int32()
        self._check(self.L.lzgpu_reduce_to_chain(a, _ptr(anchors), C.c_uint32(len(anchors)), C.byref(kept), C.byref(n), C.byref(best)),
                    "lzgpu_reduce_to_chain")
        out = np.zeros(n.value, dtype=np.uint32)
        if n.value:
            C.memmove(out.ctypes.data, kept, 4 * n.value)
        self.L.lzgpu_free(kept)
        return out, best.value

    def reduce_to_chain_batch(self, anchor_sets, chain_diag=0, chain_anti=0, scale=100, overlap_sub=91, diag_pen=None, anti_pen=None):
        """several (query, strand) problems at once, one host thread each -> [(kept indices, chain score)] as reduce_to_chain gives them"""
        sets = [np.ascontiguousarray(x, dtype=SEG_DTYPE) for x in anchor_sets]
        k = len(sets)
        a = (C.c_int32 * 6)(chain_diag if diag_pen is None else diag_pen, chain_anti if anti_pen is None else anti_pen,
                            chain_diag, chain_anti, scale, overlap_sub)
        ptrs = (C.c_void_p * k)(*[x.ctypes.data for x in sets]); ns = (C.c_uint32 * k)(*[len(x) for x in sets])
        kept = (C.c_void_p * k)(); nk = (C.c_uint32 * k)(); best = (C.c_int32 * k)()
        self._check(self.L.lzgpu_reduce_to_chain_batch(a, ptrs, ns, C.c_uint32(k), kept, nk, best), "lzgpu_reduce_to_chain_batch")
        res = []
        for j in range(k):
            out = np.zeros(nk[j], dtype=np.uint32)
            if nk[j]:
                C.memmove(out.ctypes.data, kept[j], 4 * nk[j])
            self.L.lzgpu_free(kept[j])
            res.append((out, best[j]))
        return res

    # ---- B1 + B2 of many windows (N3)
    def window_search(self, masked_sub, windows, sd, ctb, q=None, slot=-1, xdrop=910, hsp_threshold=3000):
        """windows: [(t_off, t_len, q_off, q_len)] on the resident target and the query -> [HSP array per window]"""
        a = WindowSearchArgs()
        sub = np.ascontiguousarray(masked_sub, dtype=np.int32)
        w = np.ascontiguousarray(np.array(windows, dtype=np.uint32).reshape(-1, 4))
        ctb = np.ascontiguousarray(ctb, dtype=np.int8)
        if q is not None:
            q = np.ascontiguousarray(q, dtype=np.uint8)
            a.query, a.qlen = q.ctypes.data, len(q)
        else:
            a.query, a.qlen = None, self._keep[("qlen", slot)]
        a.query_slot = slot
        a.sub, a.xdrop, a.hsp_threshold = sub.ctypes.data, xdrop, hsp_threshold
        a.seed, a.char_to_bits = C.addressof(sd), ctb.ctypes.data
        a.windows, a.n_windows = w.ctypes.data, len(w)
        out = C.c_void_p(); n = C.c_uint64(); cnt = C.c_void_p()
        self._check(self._f("window_search")(C.byref(a), C.byref(out), C.byref(n), C.byref(cnt)), "lzgpu_window_search")
        hs = np.zeros(n.value, dtype=HSP_DTYPE); cn = np.zeros(len(w), dtype=np.uint32)
        if n.value:
            C.memmove(_ptr(hs), out, n.value * HSP_DTYPE.itemsize)
        if len(w):
            C.memmove(_ptr(cn), cnt, len(w) * 4)
        self.L.lzgpu_free(out); self.L.lzgpu_free(cnt)
        offs = np.concatenate([[0], np.cumsum(cn.astype(np.int64))]).astype(np.int64)
        return [hs[offs[k]:offs[k + 1]] for k in range(len(w))]

    # ---- instrumentation
    def counters_reset(self):
        self._f("counters_reset")()

    def counters(self):
        c = Counters()
        self._f("counters_get")(C.byref(c))
        return {k: int(getattr(c, k)) for k, _ in Counters._fields_}

    def set_hit_capacity(self, n):
        self._check(self._f("set_hit_capacity")(n), "set_hit_capacity")

    def set_dp_slot(self, nbytes):
        self._check(self.L.lzgpu_set_dp_slot(nbytes), "set_dp_slot")

    def set_dp_window(self, n):
        self._check(self.L.lzgpu_set_dp_window(n), "set_dp_window")

    def dp_longest(self, reset=False):
        """{rows, cells, sweep_ticks, traceback_ticks} of the DP that swept the most rows since the last reset"""
        out = (C.c_uint64 * 4)()
        self.L.lzgpu_dp_longest(out, int(reset))
        return dict(zip(("rows", "cells", "sweep_ticks", "traceback_ticks"), (int(v) for v in out)))

    def profile_enable(self, on=True):
        self.L.lzgpu_profile_enable(int(on))

    def profile_reset(self):
        self.L.lzgpu_profile_reset()

    def profile(self):
        out, i = {}, 0
        while True:
            name = C.c_char_p(); n = C.c_uint64(); ms = C.c_double()
            if self.L.lzgpu_profile_get(i, C.byref(name), C.byref(n), C.byref(ms)):
                break
            out[name.value.decode()] = {"launches": int(n.value), "ms": float(ms.value)}
            i += 1
        return out
