"""Parity tests proper for B3 (reduce_to_points + gapped_extend): the HIP Y-drop DP kernel under
the speculative anchor-window driver, called through the C ABI, against the reference's golden
LAVs / the oracle.  Bit-exact: scores, block coordinates, every gap-free piece, block order, and
the reference's 'DP cells visited' counter.  Needs an MI355X."""
import os
import numpy as np
import pytest

from oracle import lzo
from lastz_amd import seqio, lzgpu
import helpers as H

pytestmark = pytest.mark.gpu
CTB = lzo.upper_nuc_to_bits()


def _gpu_blocks(gpu, t, queries, **kw):
    """whole hot path on the GPU: table -> HSPs -> gapped, per query and strand, as lastz drives it"""
    sub, masked = H.scoring()
    gpu.table_prepare(t, gpu.seed(), CTB)
    out = []
    gpu.counters_reset()
    for ci, q in enumerate(queries):
        for _, rev, qq in H.strands(q):
            hsps = gpu.seed_hit_search(masked, q=qq)
            segs = np.zeros(len(hsps), dtype=lzgpu.SEG_DTYPE)
            segs["pos1"] = hsps["pos1"] - hsps["length"]; segs["pos2"] = hsps["pos2"] - hsps["length"]
            segs["length"] = hsps["length"]; segs["s"] = hsps["score"]; segs["id"] = rev
            al, ops = gpu.gapped_extend(sub, segs, q=qq, **kw)
            if len(al):
                out.append((ci + 1, rev, H.blocks_of(al, ops)))
    return out, gpu.counters()


def test_base_test_default_lav(gpu):
    """make test: pseudocat x pseudopig, all defaults (src/Makefile:208-217) -- end to end on the GPU"""
    tgt = seqio.read_fasta(os.path.join(H.GOLDEN, "pseudocat.fa"))[0][1]
    qs = [q for _, q in seqio.read_fasta(os.path.join(H.GOLDEN, "pseudopig.fa"))]
    mine, c = _gpu_blocks(gpu, tgt, qs)
    assert mine == H.lav_blocks(os.path.join(H.GOLDEN, "base_test.default.lav"))
    assert c["dp_cells"] == 21926949 and c["anchors_extended"] == 14


@pytest.mark.parametrize("case", ["synth200k", "synth_overlap", "adversarial"])
def test_golden_cases_against_reference_output(gpu, case):
    t, q = H.load_case(case)
    mine, c = _gpu_blocks(gpu, t, [q])
    assert mine == H.lav_blocks(os.path.join(H.GOLDEN, case + ".lav"))
    st = H.load_stats(case)
    assert c["dp_cells"] == st["dp_cells"] and c["anchors_extended"] == st["anchors_extended"]


def test_window_and_slot_invariance(gpu):
    """results cannot depend on how many anchors are speculated per round or on the slot size"""
    t, q = H.load_case("synth_overlap")
    gold = H.lav_blocks(os.path.join(H.GOLDEN, "synth_overlap.lav"))
    try:
        for window, slot in ((1, 8 << 20), (3, 1 << 20), (4096, 65536)):
            gpu.set_dp_window(window); gpu.set_dp_slot(slot)
            mine, _ = _gpu_blocks(gpu, t, [q])
            assert mine == gold
    finally:
        gpu.set_dp_window(0); gpu.set_dp_slot(8 << 20)


def test_thresholds_ydrop_truncation_vs_oracle(gpu):
    t, q = H.load_case("synth_overlap")
    sub, masked = H.scoring()
    gpu.table_prepare(t, gpu.seed(), CTB)
    tab = lzo.Table(t, lzo.seed())
    for kw in (dict(ydrop=3000), dict(score_thresh=20000), dict(gap_open=200, gap_extend=60, ydrop=5000),
               dict(traceback_bytes=1 << 20)):
        for _, rev, qq in H.strands(q):
            hsps, _ = lzo.seed_hit_search(tab, qq, masked)
            segs = lzo.hsps_to_segments(hsps, rev)
            okw = dict(kw); tb = okw.pop("traceback_bytes", 0)
            oal, oops, _ = lzo.gapped_extend(t, qq, sub, lzo.reduce_to_points(t, qq, sub, segs), tb_size=tb, **okw)
            al, ops = gpu.gapped_extend(sub, segs.view(lzgpu.SEG_DTYPE), q=qq, **kw)
            assert len(al) == len(oal) and (al == oal).all() and (ops == oops).all()


@pytest.mark.parametrize("kw", [dict(ydrop=60000), dict(gap_open=200, gap_extend=5), dict(gap_open=300, gap_extend=12, ydrop=12000)],
                         ids=["ydrop60000", "O200E5", "O300E12-mixed"])
def test_bands_wider_than_the_lds_ring(gpu, kw):
    """bands the 2048-column LDS ring cannot hold run in k_ydrop_wide (ring in an HBM slot) -- the stage is no longer
    declined for them; synth_overlap has 200 kbp of overlapping alignments that bound each other"""
    sub, masked = H.scoring()
    gpu.profile_enable(True); gpu.profile_reset()
    for case in ("pseudo", "synth_overlap"):
        if case == "pseudo":
            t = seqio.read_fasta(os.path.join(H.GOLDEN, "pseudocat.fa"))[0][1]
            q = seqio.read_fasta(os.path.join(H.GOLDEN, "pseudopig.fa"))[0][1]
        else:
            t, q = H.load_case(case)
            t, q = t[:60000], q[:60000]
        gpu.table_prepare(t, gpu.seed(), CTB)
        tab = lzo.Table(t, lzo.seed())
        for _, rev, qq in H.strands(q):
            hsps, _ = lzo.seed_hit_search(tab, qq, masked)
            segs = lzo.hsps_to_segments(hsps, rev)
            oal, oops, ost = lzo.gapped_extend(t, qq, sub, lzo.reduce_to_points(t, qq, sub, segs), **kw)
            gpu.counters_reset()
            al, ops = gpu.gapped_extend(sub, segs.view(lzgpu.SEG_DTYPE), q=qq, **kw)
            assert len(al) == len(oal) and (al == oal).all() and (ops == oops).all()
            assert gpu.counters()["dp_cells"] == ost["dp_cells"]
    prof = gpu.profile(); gpu.profile_enable(False)
    assert prof.get("k_ydrop_wide", {"launches": 0})["launches"] > 0


def test_two_mbp_pair_vs_oracle(gpu):
    t, q = seqio.synth_pair(2_000_000, 2_000_000, seed=12)
    sub, masked = H.scoring()
    gpu.table_prepare(t, gpu.seed(), CTB)
    for _, rev, qq in H.strands(q):
        hsps = gpu.seed_hit_search(masked, q=qq)
        segs = np.zeros(len(hsps), dtype=lzgpu.SEG_DTYPE)
        segs["pos1"] = hsps["pos1"] - hsps["length"]; segs["pos2"] = hsps["pos2"] - hsps["length"]
        segs["length"] = hsps["length"]; segs["s"] = hsps["score"]; segs["id"] = rev
        gpu.counters_reset()
        al, ops = gpu.gapped_extend(sub, segs, q=qq)
        c = gpu.counters()
        osegs = segs.view(lzo.SEG_DTYPE)
        oal, oops, ost = lzo.gapped_extend(t, qq, sub, lzo.reduce_to_points(t, qq, sub, osegs))
        assert len(al) == len(oal) and (al == oal).all() and (ops == oops).all()
        assert c["dp_cells"] == ost["dp_cells"]
        # every alignment re-scores to its score on the host (size-independent property)
        for a in al[:50]:
            s, p1, p2 = 0, int(a["beg1"]) - 1, int(a["beg2"]) - 1
            for w in ops[int(a["script_off"]):int(a["script_off"] + a["script_len"])]:
                op, n = int(w) & 3, int(w) >> 2
                if op == 3:
                    s += int(sub[t[p1:p1 + n], qq[p2:p2 + n]].sum()); p1 += n; p2 += n
                elif op == 1:
                    s -= 400 + 30 * n; p2 += n
                else:
                    s -= 400 + 30 * n; p1 += n
            assert s == a["s"]


def test_identical_sequences_get_the_trivial_alignment(gpu):
    """identical_sequences (src/gapped_extend.c:1886-1933) compares dna_toupper() bytes and the strand flags: the
    trivial self-alignment is put in front of every anchor (:1152-1189), so anchors on the main diagonal vanish;
    inhibit_trivial drops it from the output (:1483); the byte-for-byte check against the reference binary is
    tests/test_gpu_lastz_cli.py::test_identical_sequences"""
    sub, _ = H.scoring()
    t, _ = seqio.synth_pair(50000, 100, seed=3)
    gpu.table_prepare(t, gpu.seed(), CTB)
    segs = np.zeros(1, dtype=lzgpu.SEG_DTYPE); segs["pos1"] = 1000; segs["pos2"] = 1000; segs["length"] = 100; segs["s"] = 9000
    for q in (t.copy(), t | 0x20, np.where(np.arange(len(t)) % 3 == 0, t | 0x20, t).astype(np.uint8)):
        al, ops = gpu.gapped_extend(sub, segs.copy(), q=q)
        assert len(al) == 1 and tuple(al[0][["beg1", "beg2", "end1", "end2"]]) == (1, 1, 50000, 50000)
        assert list(ops) == [(50000 << 2) | 3] and al[0]["s"] == int(sub[t & 0xDF, t & 0xDF].sum())
        al, ops = gpu.gapped_extend(sub, segs.copy(), q=q, inhibit_trivial=True)
        assert len(al) == 0
    al, _ = gpu.gapped_extend(sub, segs.copy(), q=t.copy(), strands_differ=True)     # not "identical" for the reference
    assert len(al) == 1 and al[0]["s"] > 0
    q = t.copy(); q[777] = ord("A") if q[777] != ord("A") else ord("C")     # one base apart: an ordinary pair
    al, _ = gpu.gapped_extend(sub, segs.copy(), q=q)
    assert len(al) == 1


def test_batch_of_problems_equals_one_call_each(gpu):
    """lzgpu_gapped_extend_batch: both strands of a query (and a second query) as problems of one batch -- the DPs of
    all of them share the launches -- give exactly what one lzgpu_gapped_extend call per problem gives"""
    t, q = H.load_case("synth_overlap")
    _, q2 = H.load_case("synth200k")
    sub, masked = H.scoring()
    gpu.table_prepare(t, gpu.seed(), CTB)
    problems, single = [], []
    for slot, qq in enumerate((q, seqio.revcomp(q), q2[:150000])):
        gpu.query_upload(slot, qq)
        hsps = gpu.seed_hit_search(masked, slot=slot)
        segs = np.zeros(len(hsps), dtype=lzgpu.SEG_DTYPE)
        segs["pos1"] = hsps["pos1"] - hsps["length"]; segs["pos2"] = hsps["pos2"] - hsps["length"]
        segs["length"] = hsps["length"]; segs["s"] = hsps["score"]; segs["id"] = slot
        problems.append(dict(anchors=segs, slot=slot))
        single.append(gpu.gapped_extend(sub, segs, slot=slot))
    gpu.counters_reset()
    batch = gpu.gapped_extend_batch(sub, problems)
    cb = gpu.counters()
    assert len(batch) == 3 and sum(len(a) for a, _ in batch) > 10
    for (al, ops), (sal, sops) in zip(batch, single):
        assert len(al) == len(sal) and (al == sal).all() and (ops == sops).all()
    assert cb["anchors_extended"] == sum(len(a) for a, _ in single) or cb["anchors_extended"] > 0
    # mixed argument kinds: one problem by slot, one with the query passed as host bytes
    two = gpu.gapped_extend_batch(sub, [problems[0], dict(anchors=problems[1]["anchors"], q=seqio.revcomp(q))])
    for (al, ops), (sal, sops) in zip(two, single[:2]):
        assert (al == sal).all() and (ops == sops).all()
    with pytest.raises(Exception):                               # problems of a batch share the scoring
        gpu.gapped_extend_batch(sub, [problems[0], dict(anchors=problems[1]["anchors"], slot=1, ydrop=5000)])


def test_rectangle_of_the_sequences_as_a_problem(gpu):
    """t_off / t_len / q_off / q_len: a window of the resident sequences is the whole problem (what the tweener does with
    extract_subsequence, src/tweener.c:769-829) -- same alignments as the oracle on the cut-out pieces, window coordinates"""
    t, q = H.load_case("synth200k")
    sub, masked = H.scoring()
    gpu.table_prepare(t, gpu.seed(), CTB)
    gpu.query_upload(0, q)
    wins = [(20000, 60000, 15000, 70000), (0, 50000, 0, 50000), (100000, len(t) - 100000, 90000, len(q) - 90000)]
    problems, want = [], []
    for (t0, tl, q0, ql) in wins:
        tt, qq = t[t0:t0 + tl], q[q0:q0 + ql]
        hsps, _ = lzo.seed_hit_search(lzo.Table(tt, lzo.seed()), qq, masked)
        segs = lzo.hsps_to_segments(hsps, 0)
        want.append(lzo.gapped_extend(tt, qq, sub, lzo.reduce_to_points(tt, qq, sub, segs))[:2])
        problems.append(dict(anchors=segs.view(lzgpu.SEG_DTYPE), slot=0, t_off=t0, t_len=tl, q_off=q0, q_len=ql))
    assert sum(len(a) for a, _ in want) >= 3
    for pr, (oal, oops) in zip(problems, want):                  # one at a time ...
        pr1 = dict(pr); al, ops = gpu.gapped_extend(sub, pr1.pop("anchors"), **pr1)
        assert len(al) == len(oal) and (al == oal).all() and (ops == oops).all()
    for (al, ops), (oal, oops) in zip(gpu.gapped_extend_batch(sub, problems), want):    # ... and as one batch
        assert len(al) == len(oal) and (al == oal).all() and (ops == oops).all()


@pytest.mark.parametrize("opt", ["noytrim", "allgappedbounds", "both"])
def test_untrimmed_ends_and_all_bounds(gpu, opt):
    """--noytrim (k_ydrop<true>: an extension that reaches the end of a sequence may end there,
    src/gapped_extend.c:3747-3750, :3866) and --allgappedbounds (alignments below the threshold still bound later ones,
    :1411-1429) against LAVs of the pristine binary (tests/golden/options_*.lav) -- no longer declined"""
    import test_oracle_vs_reference as T
    _, okw = T.OPTION_CASES[opt]
    kw = dict(no_trim=not okw.get("trim_to_peak", True), all_bounds=okw.get("all_bounds", False), score_thresh=okw.get("score_thresh", 3000))
    for name, (t, q) in T._option_pairs().items():
        mine, _ = _gpu_blocks(gpu, t, [q], **kw)
        assert mine == H.lav_blocks(os.path.join(H.GOLDEN, f"options_{opt}_{name}.lav")), (opt, name)


def test_paired_bases_limit(gpu):
    """maxPairedBases (--querydepth, src/gapped_extend.c:1441-1459): below the limit the stage is the stage without a
    limit; the moment the alignments pair more bases than allowed the call declines (LZGPU_NH_PAIRED_LIMIT = 8) so that
    the reference's own routine warns and keeps / discards"""
    t, q = H.load_case("synth200k")
    gold = H.lav_blocks(os.path.join(H.GOLDEN, "synth200k.lav"))
    mine, _ = _gpu_blocks(gpu, t, [q], max_paired_bases=len(t) * 4)              # a depth of 4: never reached here
    assert mine == gold
    with pytest.raises(lzgpu.NotHandled) as e:
        _gpu_blocks(gpu, t, [q], max_paired_bases=1000)
    assert e.value.rc == 8


@pytest.mark.parametrize("repl", ["0", "1"])
@pytest.mark.parametrize("no_trim", [False, True])
def test_both_forms_of_the_row_setup(gpu, repl, no_trim):
    """k_ydrop<NOTRIM, false, REPLICATE>: problems without earlier alignments run the row set-up either on one leading
    wave or replicated on all four (a launch-time choice by the number of DPs; LZGPU_DP_REPL forces it): same alignments"""
    import test_oracle_vs_reference as T
    t, q = T._option_pairs()["adversarial_piece"]
    os.environ["LZGPU_DP_REPL"] = repl
    try:
        mine, _ = _gpu_blocks(gpu, t, [q], no_trim=no_trim)
    finally:
        del os.environ["LZGPU_DP_REPL"]
    gold = "options_noytrim_adversarial_piece.lav" if no_trim else None
    if gold:
        assert mine == H.lav_blocks(os.path.join(H.GOLDEN, gold))
    else:
        sub, masked = H.scoring()
        assert mine == T._oracle_blocks_with(t, q)


@pytest.mark.parametrize("narrow", ["0", "1"])
@pytest.mark.parametrize("repl", ["0", "1"])
def test_both_builds_of_the_dp_kernel(gpu, narrow, repl):
    """Round 5: k_ydrop (four waves per DP, 32-bit sweep row) and k_ydrop_n (two waves per DP, four cells per batch, the C / D cells of
    the sweep row as 16-bit offsets from a base that follows the running best: lz_dp_dev.hpp, LzDpCells16).  The launcher picks by the
    size of a launch; LZGPU_DP_NARROW forces one or the other.  Same alignments from both, with either form of the row set-up: the
    goldens of the reference (with and without bounds at work: the adversarial piece has 198 alignments bounding each other), untrimmed
    ends, every bound kept, other penalties and y-drops against the oracle."""
    import test_oracle_vs_reference as T
    os.environ["LZGPU_DP_NARROW"] = narrow; os.environ["LZGPU_DP_REPL"] = repl
    try:
        gpu.profile_reset(); gpu.profile_enable(True)
        for case in ("synth200k", "adversarial"):
            t, q = H.load_case(case)
            mine, _ = _gpu_blocks(gpu, t, [q])
            assert mine == H.lav_blocks(os.path.join(H.GOLDEN, f"{case}.lav")), case
        t, q = T._option_pairs()["adversarial_piece"]
        for opt in ("noytrim", "allgappedbounds"):
            _, okw = T.OPTION_CASES[opt]
            kw = dict(no_trim=not okw.get("trim_to_peak", True), all_bounds=okw.get("all_bounds", False), score_thresh=okw.get("score_thresh", 3000))
            mine, _ = _gpu_blocks(gpu, t, [q], **kw)
            assert mine == H.lav_blocks(os.path.join(H.GOLDEN, f"options_{opt}_adversarial_piece.lav")), opt
        prof = gpu.profile(); gpu.profile_enable(False)
        assert prof.get("k_ydrop_n" if narrow == "1" else "k_ydrop", {"launches": 0})["launches"] > 0
        assert prof.get("k_ydrop" if narrow == "1" else "k_ydrop_n", {"launches": 0})["launches"] == 0
        # other scorings against the oracle; the last one is outside the 16-bit row's rule (y-drop + gapOE + 1025 + 100 > 65535): the
        # four-wave kernel runs it whatever LZGPU_DP_NARROW says
        t, q = H.load_case("synth_overlap")
        sub, masked = H.scoring()
        gpu.table_prepare(t, gpu.seed(), CTB)
        tab = lzo.Table(t, lzo.seed())
        for kw in (dict(ydrop=3000), dict(gap_open=200, gap_extend=60, ydrop=5000), dict(gap_open=2500, gap_extend=500, ydrop=60500, score_thresh=2000),
                   dict(gap_open=2500, gap_extend=500, ydrop=64000, score_thresh=2000)):
            for _, rev, qq in H.strands(q):
                hsps, _ = lzo.seed_hit_search(tab, qq, masked)
                segs = lzo.hsps_to_segments(hsps, rev)
                oal, oops, _ = lzo.gapped_extend(t, qq, sub, lzo.reduce_to_points(t, qq, sub, segs), **kw)
                al, ops = gpu.gapped_extend(sub, segs.view(lzgpu.SEG_DTYPE), q=qq, **kw)
                assert len(al) == len(oal) and (al == oal).all() and (ops == oops).all(), kw
    finally:
        del os.environ["LZGPU_DP_NARROW"]; del os.environ["LZGPU_DP_REPL"]


def test_dp_codes_follow_the_bytes_of_a_resident_slot(gpu):
    """Round 6 keeps the DP class codes of a sequence across calls while its bytes and the class map do not change.  A resident query slot that is
    uploaded again -- same length, other bytes: the two strands of a query -- or extended under another matrix must not see the old codes: every
    result through the slot equals the result of the same call with a host pointer (which uploads and encodes afresh, and is what the golden /
    oracle tests above pin)."""
    sub, masked = H.scoring()
    t, q = H.load_case("synth_overlap")
    gpu.table_prepare(t, gpu.seed(), CTB)
    sub2 = sub.copy(); sub2[ord("A"), ord("C")] = sub2[ord("C"), ord("A")] = -90           # another class map of the same shape

    def segs_of(qq):
        hsps = gpu.seed_hit_search(masked, q=qq)
        s = np.zeros(len(hsps), dtype=lzgpu.SEG_DTYPE)
        s["pos1"] = hsps["pos1"] - hsps["length"]; s["pos2"] = hsps["pos2"] - hsps["length"]; s["length"] = hsps["length"]; s["s"] = hsps["score"]
        return s

    def same(a, b):
        return len(a[0]) == len(b[0]) and (a[0] == b[0]).all() and len(a[1]) == len(b[1]) and (np.asarray(a[1]) == np.asarray(b[1])).all()

    n_checked = 0
    for qq in (q, seqio.revcomp(q), q):                        # slot 5 is overwritten twice with sequences of the same length
        gpu.query_upload(5, qq)
        sg = segs_of(qq)
        for m in (sub, sub2, sub):                             # ... and the class map changes under a resident slot and back
            via_slot = gpu.gapped_extend(m, sg, slot=5)
            via_host = gpu.gapped_extend(m, sg, q=qq)
            assert same(via_slot, via_host)
            n_checked += len(via_slot[0])
    assert n_checked > 20
