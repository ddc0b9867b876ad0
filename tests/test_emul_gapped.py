"""CPU check of the B3 design: the product's one-sided DP (lz_dp_dev.hpp: the exact code the
gfx950 kernel runs, one wave per DP, three parallel walks per row) executed lane by lane / phase
by phase, under the product's host orchestration (speculative anchor windows, commit-in-order
validation).  Must reproduce the oracle's alignments, edit scripts and DP-cell counts bit for bit."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest

from oracle import lzo
from lastz_amd import seqio, lzgpu
import helpers as H

EMUL_DIR = os.path.join(H.ROOT, "tests", "emul")
CS = os.path.join(H.ROOT, "lastz_amd", "csrc")


ARGTYPES = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_void_p, C.c_uint32,
            C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]


@pytest.fixture(scope="module")
def L():
    so = H.build_emul()
    lib = C.CDLL(so)
    lib.emul_gapped_extend.argtypes = ARGTYPES
    return lib


def emul_gapped(L, t, q, sub, segs, window=1024, tb_slot=0, ydrop=9400, thresh=3000, tb_len=0, gap_open=400, gap_extend=30,
                all_bounds=False, no_trim=False):
    L.emul_gapped_options(int(all_bounds), int(no_trim))
    t = np.ascontiguousarray(np.append(t, 0).astype(np.uint8)); q = np.ascontiguousarray(np.append(q, 0).astype(np.uint8))
    segs = np.ascontiguousarray(segs.copy())
    out = C.c_void_p(); n = C.c_uint64(); ops = C.c_void_p(); nops = C.c_uint64()
    rc = L.emul_gapped_extend(t.ctypes.data, len(t) - 1, q.ctypes.data, len(q) - 1, sub.ctypes.data, gap_open, gap_extend, ydrop, thresh,
                              tb_len, segs.ctypes.data, len(segs), 1, window, tb_slot,
                              C.byref(out), C.byref(n), C.byref(ops), C.byref(nops))
    assert rc == 0, rc
    al = np.zeros(n.value, dtype=lzgpu.ALIGN_DTYPE); op = np.zeros(nops.value, dtype=np.uint32)
    if n.value:
        C.memmove(al.ctypes.data, out, n.value * al.itemsize)
    if nops.value:
        C.memmove(op.ctypes.data, ops, nops.value * 4)
    st = (C.c_uint64 * 8)(); L.emul_gapped_stats(st)
    return al, op, dict(zip(("anchors", "anchors_extended", "dp_runs", "dp_cells", "rounds", "reruns", "retries", "wide_runs"), st))


def _check(L, t, q, **kw):
    sub, masked = H.scoring()
    tab = lzo.Table(t, lzo.seed())
    tot = {}
    for _, rev, qq in H.strands(q):
        hsps, _ = lzo.seed_hit_search(tab, qq, masked)
        segs = lzo.hsps_to_segments(hsps, rev)
        oal, oops, ost = lzo.gapped_extend(t, qq, sub, lzo.reduce_to_points(t, qq, sub, segs), ydrop=kw.get("ydrop", 9400),
                                           tb_size=kw.get("tb_len", 0), gap_open=kw.get("gap_open", 400), gap_extend=kw.get("gap_extend", 30),
                                           score_thresh=kw.get("thresh", 3000), all_bounds=kw.get("all_bounds", False),
                                           trim_to_peak=not kw.get("no_trim", False))
        eal, eops, est = emul_gapped(L, t, qq, sub, segs.view(lzgpu.SEG_DTYPE), **kw)
        assert len(oal) == len(eal) and (oal == eal).all() and (oops == eops).all()
        assert est["dp_cells"] == ost["dp_cells"] and est["anchors_extended"] == ost["anchors_extended"]
        for k, v in est.items():
            tot[k] = tot.get(k, 0) + v
    return tot


def test_reference_inputs(L):
    tgt = seqio.read_fasta(os.path.join(H.GOLDEN, "pseudocat.fa"))[0][1]
    for _, q in seqio.read_fasta(os.path.join(H.GOLDEN, "pseudopig.fa")):
        _check(L, tgt, q)
    _, q = seqio.read_fasta(os.path.join(H.GOLDEN, "pseudopig.fa"))[0]
    _check(L, tgt, q, window=1)                       # one anchor per round == the reference's serial loop
    st = _check(L, tgt, q, tb_slot=65536)             # tiny first-try slots: overflow -> re-run in bigger ones
    assert st["retries"] > 0


@pytest.mark.parametrize("case", ["synth200k", "synth_overlap"])
def test_golden_cases(L, case):
    t, q = H.load_case(case)
    _check(L, t, q)


def test_obstacle_course(L):
    """tandem repeats: hundreds of overlapping alignments bound / mask each other (L,R bounds, active
    segments, window cuts and re-runs)"""
    t, q = H.load_case("adversarial")
    st = _check(L, t[9000:14500], q[29500:34000])
    assert st["rounds"] > 3


def test_helper_threads_of_the_commit_pass():
    """the fork-join helpers (alignments pre-built, deferred anchors checked against their slot's alignment on several
    threads) only start for problems of >= 20 k anchors: here forced on, with chunks of one item, in a process of its own"""
    import subprocess, sys
    env = dict(os.environ); env["LZGPU_HELPER_MIN_ANCHORS"] = "0"
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", os.path.join(H.ROOT, "tests", "test_emul_gapped.py"),
                        "-k", "golden_cases or obstacle_course"], capture_output=True, text=True, env=env, cwd=H.ROOT, timeout=1500)
    assert p.returncode == 0 and " passed" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_a_sweep_that_passes_its_piece_horizon_is_run_again():
    """Round 5: bounds and masked cells reach the DP as pieces of rows worked out on the host (lz_dp_pieces.cpp), complete up to a
    horizon; a sweep that gets further stops (LZ_DP_PIECE_SLOT) and is run again with more pieces.  With a first horizon of 97 rows the
    obstacle course (hundreds of alignments bounding and masking each other) takes that path in most of its DPs: same alignments,
    scripts and cell counts as the oracle."""
    import subprocess, sys
    code = ("import sys, ctypes as C\n"
            "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import helpers as H, test_emul_gapped as T\n"
            "lib = C.CDLL(H.build_emul()); lib.emul_gapped_extend.argtypes = T.ARGTYPES; lib.emul_gapped_piece_reruns.restype = C.c_uint64\n"
            "t, q = H.load_case('adversarial')\n"
            "st = T._check(lib, t[9000:14500], q[29500:34000])\n"
            "t, q = H.load_case('synth_overlap')\n"
            "T._check(lib, t, q)\n"
            "n = lib.emul_gapped_piece_reruns()\n"
            "assert n > 200, n\n"
            "print('reruns', n)\n" % (H.ROOT, os.path.join(H.ROOT, "tests")))
    env = dict(os.environ); env["EMUL_DP_HORIZON"] = "97"
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=H.ROOT, timeout=1500)
    assert p.returncode == 0 and "reruns" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_pieces_equal_the_references_row_by_row_bookkeeping():
    """lz_dp_pieces.cpp evaluates a DP's bounds and masked cells segment by segment; tests/emul/emul_bounds_plain.cpp restates the
    reference's routines as they stand (update_LR_bounds, next / prev_sweep_seg, update_active_segs, build_active_seg: one call per
    row).  Every job of the obstacle course and of the overlapping-alignment cases -- forward and backward sweeps, bounds that hop and
    die, alignments entering the list -- row by row, with the pieces asked for up to 61 rows, 776 rows and the job's last row."""
    import subprocess, sys
    code = ("import sys, ctypes as C\n"
            "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import helpers as H, test_emul_gapped as T\n"
            "lib = C.CDLL(H.build_emul()); lib.emul_gapped_extend.argtypes = T.ARGTYPES; lib.emul_gapped_pieces_verified.restype = C.c_uint64\n"
            "t, q = H.load_case('adversarial')\n"
            "T._check(lib, t[9000:14500], q[29500:34000])\n"
            "t, q = H.load_case('synth_overlap')\n"
            "for kw in (dict(), dict(gap_open=200, gap_extend=60, ydrop=5000), dict(tb_len=1 << 20)):\n"
            "    T._check(lib, t, q, **kw)\n"
            "n = lib.emul_gapped_pieces_verified()\n"
            "assert n > 1500, n\n"
            "print('verified', n)\n" % (H.ROOT, os.path.join(H.ROOT, "tests")))
    env = dict(os.environ); env["EMUL_VERIFY_PIECES"] = "1"
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=H.ROOT, timeout=2400)
    assert p.returncode == 0 and "verified" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_indexed_neighbour_search_equals_plain_walk(L):
    """msp_left_right with the running-max index vs the reference's walk, random snapshots (incl. > 64 overlaps)"""
    L.emul_selftest_neighbours.restype = C.c_int
    for seed, na in ((1, 0), (2, 1), (3, 40), (4, 600), (5, 3000), (6, 25000)):
        assert L.emul_selftest_neighbours(C.c_uint32(seed), C.c_uint32(na), C.c_uint32(20000)) == 0


def test_traceback_truncation_rule(L):
    """the reference truncates an alignment when its traceback arena runs out (:3640-3661)"""
    t, q = H.load_case("synth_overlap")
    _check(L, t, q, tb_len=1 << 20)


@pytest.mark.parametrize("kw", [dict(ydrop=60000), dict(gap_open=200, gap_extend=5), dict(gap_open=300, gap_extend=12, ydrop=12000)],
                         ids=["ydrop60000", "O200E5", "O300E12-mixed"])
def test_bands_wider_than_the_lds_ring(L, kw):
    """bands the 2048-column LDS ring cannot hold re-run in the HBM-ring variant of the same DP (k_ydrop_wide) instead
    of declining the stage: huge y-drops, small gap-extension penalties"""
    tgt = seqio.read_fasta(os.path.join(H.GOLDEN, "pseudocat.fa"))[0][1]
    _, q = seqio.read_fasta(os.path.join(H.GOLDEN, "pseudopig.fa"))[0]
    st = _check(L, tgt, q, **kw)
    assert st["wide_runs"] > 0


@pytest.mark.parametrize("seed", [101, 202])
def test_speculation_windows_on_random_obstacle_courses(L, seed):
    """the acceptance rule of the speculative windows (a result is kept iff the reference would have run the same DP:
    same neighbours and nothing committed since touches what it explored, or nothing at all touches it) against
    the oracle on random pairs crowded with tandem repeats and overlapping homology, for window sizes from the
    reference's serial loop (1) to everything at once; every window size must also launch what it commits"""
    rng = np.random.default_rng(seed)
    t, q = seqio.synth_pair(6000, 6000, seed=seed, block_min=300, block_max=2000, homolog_frac=0.85)
    t = t.copy(); q = q.copy()
    unit = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(rng.integers(17, 41)))
    rep = np.tile(unit, 40)
    for arr, n_rep in ((t, 2), (q, 3)):
        for _ in range(n_rep):
            s0 = int(rng.integers(0, len(arr) - len(rep))); n = int(rng.integers(len(rep) // 3, len(rep)))
            arr[s0:s0 + n] = rep[:n]
    q[1000:2200] = t[3500:4700]                                  # a second copy of a target block: competing alignments
    stats = [_check(L, t, q, window=w) for w in (1, 1024)]
    assert stats[0]["rounds"] >= stats[1]["rounds"]
    assert stats[0]["anchors_extended"] == stats[1]["anchors_extended"]


def test_sixteen_bit_mask_stamps_start_over(L):
    """The sweep row's mask stamps are 16 bits in LDS and start over every LZ_DP_STAMP_PERIOD rows (65535), the ring's
    stamps being cleared at that moment.  Built with a period of 37 rows the same code starts over hundreds of times
    inside every DP of a case full of overlapping alignments (masked cells on most rows): still the oracle's alignments,
    scripts and cell counts."""
    so = H.build_emul(tag="stamp37", flags=["-DLZ_DP_STAMP_PERIOD=37"])
    lib = C.CDLL(so)
    lib.emul_gapped_extend.argtypes = L.emul_gapped_extend.argtypes
    t, q = H.load_case("synth_overlap")
    tot = _check(lib, t, q)
    assert tot["anchors_extended"] > 5
    t, q = H.load_case("adversarial")
    _check(lib, t[:30000], q[:30000])


@pytest.mark.parametrize("kw", [dict(no_trim=True), dict(all_bounds=True, thresh=9000), dict(all_bounds=True, no_trim=True, thresh=6000)],
                         ids=["noytrim", "allgappedbounds", "both"])
def test_untrimmed_ends_and_all_bounds(L, kw):
    """--noytrim (an extension that reaches the end of a sequence may end there, src/gapped_extend.c:3747-3750, :3866)
    and --allgappedbounds (alignments below the threshold still bound later ones, :1411-1429) against the oracle: short
    sequences whose homology runs into both ends, and the tandem-repeat obstacle course"""
    t, q = H.load_case("adversarial")
    _check(L, t[9000:14500], q[29500:34000], **kw)
    t2, q2 = seqio.synth_pair(5000, 4200, seed=77, block_min=900, block_max=2500, homolog_frac=0.95)
    for lo, hi in ((0, 4200), (300, 3300), (1200, 2600)):          # windows of the query: ends inside homologous blocks
        _check(L, t2, q2[lo:hi], **kw)
        _check(L, t2[lo:hi], q2, **kw)


def test_sixteen_bit_sweep_row_equals_the_oracle():
    """Round 5: the C / D cells of the sweep row as 16-bit offsets from a base that follows the running best (LzDpCells16: everything
    at or below best - yDrop - gapOE - 1 is interchangeable, everything above is kept exactly).  Forced on for every DP the launcher's
    rule admits (the other tests of this file run every other pair of DPs that way): obstacle course, overlapping alignments, untrimmed
    ends, all bounds, a small y-drop, and penalties that put the stored offsets at the top of the 16-bit range
    (yDrop + gapOE + 1025 + 100 = 64,625)."""
    import subprocess, sys
    code = ("import sys, ctypes as C\n"
            "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import helpers as H, test_emul_gapped as T\n"
            "lib = C.CDLL(H.build_emul()); lib.emul_gapped_extend.argtypes = T.ARGTYPES; lib.emul_gapped_row16_runs.restype = C.c_uint64\n"
            "t, q = H.load_case('adversarial')\n"
            "T._check(lib, t[9000:14500], q[29500:34000])\n"
            "t, q = H.load_case('synth_overlap')\n"
            "for kw in (dict(), dict(gap_open=200, gap_extend=60, ydrop=5000), dict(ydrop=700, thresh=2000), dict(no_trim=True), dict(all_bounds=True),\n"
            "           dict(gap_open=2500, gap_extend=500, ydrop=60500, thresh=2000)):\n"
            "    n0 = lib.emul_gapped_row16_runs(); st = T._check(lib, t, q, **kw); n1 = lib.emul_gapped_row16_runs()\n"
            "    assert n1 - n0 >= st['dp_runs'] and st['wide_runs'] == 0, (kw, n0, n1, st)\n"
            "print('row16', lib.emul_gapped_row16_runs())\n" % (H.ROOT, os.path.join(H.ROOT, "tests")))
    env = dict(os.environ); env["EMUL_ROW16"] = "1"
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=H.ROOT, timeout=2400)
    assert p.returncode == 0 and "row16" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_two_waves_per_dp_and_four_cells_per_batch():
    """dp_kernels_narrow.hip compiles lz_dp_run with 128 lanes per DP, four cells per batch of LDS reads and the 16-bit sweep row
    (k_ydrop_n).  The same constants here, lane by lane: obstacle course, overlapping alignments, untrimmed ends, every bound kept,
    offsets at the top of the 16-bit range."""
    import subprocess, sys
    code = ("import sys, ctypes as C\n"
            "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import helpers as H, test_emul_gapped as T\n"
            "lib = C.CDLL(H.build_emul(tag='n128', flags=['-DLZ_DP_LANES=128', '-DLZ_DP_BATCH=4'])); lib.emul_gapped_extend.argtypes = T.ARGTYPES\n"
            "lib.emul_gapped_row16_runs.restype = C.c_uint64\n"
            "t, q = H.load_case('adversarial')\n"
            "T._check(lib, t[9000:14500], q[29500:34000])\n"
            "t, q = H.load_case('synth_overlap')\n"
            "for kw in (dict(), dict(no_trim=True), dict(all_bounds=True), dict(gap_open=2500, gap_extend=500, ydrop=60500, thresh=2000)):\n"
            "    T._check(lib, t, q, **kw)\n"
            "assert lib.emul_gapped_row16_runs() > 700\n"
            "print('row16', lib.emul_gapped_row16_runs())\n" % (H.ROOT, os.path.join(H.ROOT, "tests")))
    env = dict(os.environ); env["EMUL_ROW16"] = "1"
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=H.ROOT, timeout=2400)
    assert p.returncode == 0 and "row16" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


def test_sixteen_bit_sweep_row_under_random_scorings():
    """The 16-bit row's argument (values at or below best - yDrop - gapOE - 1 are interchangeable) does not depend on the penalties: six
    scorings drawn at random inside the launcher's rule (gap open 0-2500, extend 1-400, y-drop up to 40,000, threshold 800-4000), the
    obstacle course (hundreds of alignments bounding and masking each other) against the oracle, every DP on the 16-bit row."""
    import subprocess, sys
    code = ("import sys, ctypes as C\n"
            "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, helpers as H, test_emul_gapped as T\n"
            "lib = C.CDLL(H.build_emul()); lib.emul_gapped_extend.argtypes = T.ARGTYPES; lib.emul_gapped_row16_runs.restype = C.c_uint64\n"
            "t, q = H.load_case('adversarial')\n"
            "rng = np.random.default_rng(20260928)\n"
            "for k in range(6):\n"
            "    ge = int(rng.integers(1, 400)); go = int(rng.integers(0, 2500)); yd = int(rng.integers(max(200, 4 * ge), 40000)); th = int(rng.integers(800, 4000))\n"
            "    rng.integers(0, 9000); rng.integers(0, 9000)\n"
            "    st = T._check(lib, t[9000:13500], q[29500:33500], gap_open=go, gap_extend=ge, ydrop=yd, thresh=th)\n"
            "    assert st['wide_runs'] == 0 and st['dp_runs'] > 300, st\n"
            "assert lib.emul_gapped_row16_runs() > 3000\n"
            "print('row16', lib.emul_gapped_row16_runs())\n" % (H.ROOT, os.path.join(H.ROOT, "tests")))
    env = dict(os.environ); env["EMUL_ROW16"] = "1"
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=H.ROOT, timeout=2400)
    assert p.returncode == 0 and "row16" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
