"""CPU check of the GPU pipeline's DECOMPOSITION: tests/emul runs the very per-lane functions the
kernels execute (lastz_amd/csrc/lz_common.hpp) plus the product's host pieces (lz_host.cpp),
serially, in the device pipeline's order (count/scan/fill, stable bucket partition, one lane per
bucket, host finish) and must reproduce the oracle bit for bit -- HSPs, order and work counters.
The emulation is test infrastructure, not a fallback."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest

from oracle import lzo
from lastz_amd import seqio, lzgpu
import helpers as H

EMUL_DIR = os.path.join(H.ROOT, "tests", "emul")


@pytest.fixture(scope="module")
def em():
    so = H.build_emul()
    return lzgpu.Lib(path=so, prefix="emul_")


def _check(em, t, q, pattern=H.DEFAULT_SEED, wt=1, cap=None, step=1, masked=None, **kw):
    if masked is None: _, masked = H.scoring()
    ctb = lzo.upper_nuc_to_bits()
    osd = lzo.seed(pattern, wt)
    tab = lzo.Table(t, osd, step=step)
    sd = em.seed(pattern, wt)
    assert [sd.probe_xor[i] for i in range(sd.num_probes)] == [osd.probe_xor[i] for i in range(osd.num_probes)]
    em.table_prepare(t, sd, ctb, step=step)
    ws, wp = tab.csr()
    ews = np.zeros(len(ws), dtype=np.uint32); ewp = np.zeros(max(len(wp), 1), dtype=np.uint32)
    n = em.L.emul_table_csr(C.c_void_p(ews.ctypes.data), C.c_void_p(ewp.ctypes.data))
    assert n == len(wp) and (ews == ws).all() and (ewp[:n] == wp).all()
    em.set_hit_capacity(cap or (1 << 28))
    for _, _, qq in H.strands(q):
        for mode in (0, 1):
            ho, st = lzo.seed_hit_search(tab, qq, masked, mode=mode, **kw)
            em.counters_reset()
            he = em.seed_hit_search(masked, q=qq, extend=(mode == 0), **kw)
            c = em.counters()
            assert len(ho) == len(he) and (ho == he).all()
            assert c["words"] == st["words"] and c["raw_hits"] == st["raw_hits"]
            if mode == 0:
                assert all(c[k] == st[k] for k in ("extensions", "bp_extended", "hsps"))


def test_reference_inputs(em):
    tgt = seqio.read_fasta(os.path.join(H.GOLDEN, "pseudocat.fa"))[0][1]
    for _, q in seqio.read_fasta(os.path.join(H.GOLDEN, "pseudopig.fa")):
        _check(em, tgt, q)
        _check(em, tgt, q, cap=1024)                       # many chunks: diagEnd carried across chunks
        _check(em, tgt, q, pattern="11111111", wt=0)


@pytest.mark.parametrize("case,cap", [("synth200k", 20000), ("synth_overlap", 1 << 28), ("adversarial", 5000)])
def test_golden_cases(em, case, cap):
    t, q = H.load_case(case)
    _check(em, t, q, cap=cap)


def test_seeds_steps_thresholds(em):
    t, q = seqio.synth_pair(60000, 50000, seed=31, block_min=500, block_max=4000)
    _check(em, t, q, pattern="111101101111", wt=1, step=3)
    _check(em, t, q, pattern="11111111", wt=2, hsp_threshold=2000, xdrop=500)
    _check(em, t, q, pattern="1111111111", wt=0, entropic=False, cap=4096)


def test_matrix_with_more_than_8_classes(em):
    """general (masked, 32x32-table) scan path instead of the whole-block 8x8 one"""
    t, q = seqio.synth_pair(50000, 40000, seed=77, block_min=500, block_max=3000)
    q = q.copy(); q[::53] = ord("R"); q[7::61] = ord("Y"); q[3000:3500] |= 0x20; q[11::97] = ord("N")
    m = H.many_class_scoring()
    assert len({m[i].tobytes() for i in range(256)}) > 8
    _check(em, t, q, masked=m, hsp_threshold=2200)


def test_block_scanners_agree(em):
    """masked byte scan == whole-block byte scan == 4-bit scan, block by block, on random codes"""
    em.L.emul_scan_selftest.restype = C.c_int
    assert em.L.emul_scan_selftest(C.c_uint32(12345), C.c_uint32(20000)) == 0


def test_four_thread_sort(em):
    em.L.emul_sort4_selftest.restype = C.c_int
    for n in (10, 16384, 100003):
        assert em.L.emul_sort4_selftest(C.c_uint32(n), C.c_uint32(n)) == 0


def test_byte_code_path_without_nibbles(em, monkeypatch):
    """the same searches with the 4-bit arrays switched off (the path a >= 8-class matrix takes for the windows)"""
    monkeypatch.setenv("EMUL_NO_NIBBLES", "1")
    t, q = H.load_case("synth200k")
    _check(em, t, q, cap=20000)


def test_chunk_planner_splits_inside_a_block(em):
    # a 9-mer exact seed on a low-complexity target: single query positions carry many hits
    rng = np.random.default_rng(5)
    unit = np.frombuffer(b"ACGTACGGTACC", dtype=np.uint8)
    t = np.tile(unit, 400)
    q = np.concatenate([np.tile(unit, 30), np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 500)]])
    _check(em, t, q, pattern="111111111", wt=0, cap=1024, hsp_threshold=1000)
    _, masked = H.scoring()
    em.set_hit_capacity(1024)
    sd = em.seed("111111111", 0)
    em.table_prepare(np.tile(unit, 4000), sd, lzo.upper_nuc_to_bits())
    with pytest.raises(lzgpu.NotHandled):                  # one position alone exceeds the capacity
        em.seed_hit_search(masked, q=q)


def _mode(em):
    em.L.emul_last_scan_mode.restype = C.c_int
    return em.L.emul_last_scan_mode()


def test_scan_modes_cover_lut_and_byte_code_paths(em, monkeypatch):
    """which phase-A scanner a search takes: look-up tables on 2-bit codes without (0) / with (1) special-byte
    masks, or the byte-code scans (2); every forced downgrade must give the same answer"""
    t, q = H.load_case("synth200k")
    _check(em, t, q, cap=50000)
    assert _mode(em) == 0
    ta, qa = H.load_case("adversarial")
    _check(em, ta, qa, cap=5000)
    assert _mode(em) == 1
    for forced in ("1", "2"):
        monkeypatch.setenv("EMUL_SCAN_MODE", forced)
        _check(em, t, q, cap=50000)
        assert _mode(em) == int(forced)
        _check(em, ta, qa, cap=5000)
    monkeypatch.delenv("EMUL_SCAN_MODE")
    qr = q.copy(); qr[::997] = ord("R")                    # an IUPAC byte scores fillScore (-100): not a scan terminator --
    _check(em, t, qr, cap=50000)                           # consumed with its real score, the scan goes on behind it (round 4)
    assert _mode(em) == 1


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_lut_scans_with_dense_specials_and_short_sequences(em, seed):
    """special bytes every few bases around the seeds, sequence ends inside the windows, several xDrop values:
    the window limits (hard end, special base consumed, partial groups) of lz_lut_window"""
    rng = np.random.default_rng(seed)
    t, q = seqio.synth_pair(30000, 24000, seed=100 + seed, block_min=300, block_max=2500)
    t = t.copy(); q = q.copy()
    for arr in (t, q):
        n = len(arr)
        idx = rng.integers(0, n, n // 40)
        arr[idx] |= 0x20                                     # scattered lower case
        for s in rng.integers(0, n - 50, 60):
            arr[s:s + int(rng.integers(1, 40))] = ord("N")   # N runs
    _, masked = H.scoring()
    for xd in (910, 400, 375):
        _check(em, t, q, masked=masked, xdrop=xd, hsp_threshold=2000, cap=30000)
        assert _mode(em) == 1
    _check(em, t[:300], q[:200], pattern="11111111", wt=0, hsp_threshold=800)          # everything within reach of an end
    _check(em, t, q, xdrop=374, hsp_threshold=1500)         # three bases after a maximum set inside a group can lose 375 > xDrop: not eligible
    assert _mode(em) == 2


@pytest.mark.parametrize("seed", [4, 5])
def test_lut_scans_through_special_bytes_that_do_not_end_a_scan(em, seed):
    """IUPAC bytes (fillScore -100 in lastz's matrices, partial credit in the many-class matrix) single, in pairs and in
    runs, next to the seeds and at group / window boundaries: a scan consumes them with their real scores and goes on
    behind them (a new window from the next base); lower case and N (-1000) still end it"""
    rng = np.random.default_rng(seed)
    t, q = seqio.synth_pair(40000, 30000, seed=200 + seed, block_min=300, block_max=3000)
    t = t.copy(); q = q.copy()
    iupac = np.frombuffer(b"RYKMSWBDHV", dtype=np.uint8)
    for arr in (t, q):
        n = len(arr)
        idx = rng.integers(0, n, n // 25)
        arr[idx] = iupac[rng.integers(0, len(iupac), len(idx))]          # scattered, every ~25 bases: several per window
        for s in rng.integers(0, n - 20, 40):
            k = int(rng.integers(2, 12))
            arr[s:s + k] = iupac[rng.integers(0, len(iupac), k)]         # runs: one window per base
        arr[rng.integers(0, n, n // 300)] |= 0x20                        # and a few bytes that do end a scan
    _, masked = H.scoring()
    _check(em, t, q, masked=masked, hsp_threshold=1800, cap=30000)
    assert _mode(em) == 1
    _check(em, t, q, masked=H.many_class_scoring(), hsp_threshold=1800, xdrop=600)
    assert _mode(em) == 1
    _check(em, t[:500], q[:400], pattern="11111111", wt=0, masked=masked, hsp_threshold=600)


def test_host_finish_with_many_candidates(em):
    """the threaded path of lzh_finish_hsps (>= 16384 candidates: chunk sorts, merge levels, entropy factors on up to
    16 threads) against a serial statement; ties in (query position, probe) are ordered by target position"""
    em.L.emul_finish_selftest.restype = C.c_int
    for seed, n in ((1, 16384), (2, 50001), (3, 200000)):
        assert em.L.emul_finish_selftest(seed, n) == 0
