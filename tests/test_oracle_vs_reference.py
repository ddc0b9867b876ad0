"""Pins the CPU oracle (oracle/*.c) against
  (a) the reference's own golden vectors (test_data/base_test.*.lav, copied as data fixtures) and
  (b) stage-level outputs of the pristine reference binary on seeded inputs (tests/golden/*.tsv,
      *.lav, *.stats.json, produced by tests/golden/make_golden.py).
CPU only."""
import os
import numpy as np
import pytest

from oracle import lzo
from lastz_amd import seqio
import helpers as H

G = H.GOLDEN


@pytest.fixture(scope="module")
def cat_pig():
    tgt = seqio.read_fasta(os.path.join(G, "pseudocat.fa"))[0][1]
    qs = seqio.read_fasta(os.path.join(G, "pseudopig.fa"))
    return tgt, qs


def test_default_seed_constants():
    # SURVEY.md A.2, measured from `lastz --debug=90` on the reference
    sd = lzo.seed()
    assert [(sd.shift[i], sd.mask[i]) for i in range(sd.num_parts)] == [(0, 0x00F0CCFF), (16, 0x000F3000), (28, 0x00000300)]
    assert [sd.flips[i] for i in range(sd.num_flips)] == [0x2, 0x8, 0x20, 0x80, 0x800, 0x8000, 0x200000, 0x800000,
                                                          0x2000, 0x20000, 0x80000, 0x200]
    assert sd.num_probes == 13


def test_score_constants():
    sub, masked = H.scoring()
    assert lzo.NEG_INF == -1932735283 and lzo.VERY_BAD == -107374182
    assert sub[ord("A"), ord("A")] == 91 and sub[ord("c"), ord("G")] == -125
    assert masked[ord("a"), ord("A")] == -1000 and masked[ord("N"), ord("A")] == -1000
    assert sub[ord("N"), ord("A")] == -100 and sub[0, ord("A")] == lzo.VERY_BAD


def test_base_test_hits_lav(cat_pig):
    """raw seed-hit enumeration: W=8 T=0 --plus --nogfextend --nogapped (src/Makefile:295-304)"""
    tgt, qs = cat_pig
    _, masked = H.scoring()
    gold = H.lav_blocks(os.path.join(G, "base_test.hits.lav"))
    sd = lzo.seed("11111111", 0)
    tab = lzo.Table(tgt, sd)
    assert [g[0] for g in gold] == [1, 2, 3] and all(g[1] == 0 for g in gold)
    total = 0
    for (contig, rev, blocks), (_, q) in zip(gold, qs):
        hits, _ = lzo.seed_hit_search(tab, q, masked, mode=1)
        mine = [((int(h["pos1"]) - 8 + 1, int(h["pos2"]) - 8 + 1), (int(h["pos1"]), int(h["pos2"]))) for h in hits]
        assert mine == [(b["b"], b["e"]) for b in blocks]
        total += len(mine)
    assert total == 6544


def test_base_test_hsp_lav(cat_pig):
    """X-drop HSPs, entropy, threshold, discovery order: C=3 W=8 T=0 (src/Makefile:306-315)"""
    tgt, qs = cat_pig
    _, masked = H.scoring()
    gold = H.lav_blocks(os.path.join(G, "base_test.hsp.lav"))
    sd = lzo.seed("11111111", 0)
    tab = lzo.Table(tgt, sd)
    mine = []
    for ci, (_, q) in enumerate(qs):
        for _, rev, qq in H.strands(q):
            hsps, _ = lzo.seed_hit_search(tab, qq, masked)
            if len(hsps):
                mine.append((ci + 1, rev, [{"score": int(h["score"]),
                                            "b": (int(h["pos1"] - h["length"]) + 1, int(h["pos2"] - h["length"]) + 1),
                                            "e": (int(h["pos1"]), int(h["pos2"])),
                                            "l": [(int(h["pos1"] - h["length"]) + 1, int(h["pos2"] - h["length"]) + 1,
                                                   int(h["pos1"]), int(h["pos2"]))]} for h in hsps]))
    assert mine == gold
    assert sum(len(m[2]) for m in mine) == 21


def _oracle_gapped_blocks(tgt, queries, pattern=H.DEFAULT_SEED, wt=1):
    sub, masked = H.scoring()
    sd = lzo.seed(pattern, wt)
    tab = lzo.Table(tgt, sd)
    out, cells = [], 0
    for ci, q in enumerate(queries):
        for _, rev, qq in H.strands(q):
            hsps, _ = lzo.seed_hit_search(tab, qq, masked)
            anchors = lzo.reduce_to_points(tgt, qq, sub, lzo.hsps_to_segments(hsps, rev))
            al, ops, st = lzo.gapped_extend(tgt, qq, sub, anchors)
            cells += st["dp_cells"]
            if len(al):
                out.append((ci + 1, rev, H.blocks_of(al, ops)))
    return out, cells


def test_base_test_default_lav(cat_pig):
    """the whole hot path end to end, all defaults (make test; src/Makefile:208-217,329-338)"""
    tgt, qs = cat_pig
    mine, cells = _oracle_gapped_blocks(tgt, [q for _, q in qs])
    assert mine == H.lav_blocks(os.path.join(G, "base_test.default.lav"))
    assert sum(len(m[2]) for m in mine) == 14
    assert cells == 21926949            # "DP cells visited" of the reference's collect_stats build


@pytest.mark.parametrize("case", ["synth200k", "synth_overlap", "adversarial"])
def test_hsp_stage_vs_reference_output(case):
    t, q = H.load_case(case)
    _, masked = H.scoring()
    gold = H.read_hsp_tsv(os.path.join(G, case + ".hsp.tsv"))
    gst = H.load_stats(case)
    sd = lzo.seed()
    tab = lzo.Table(t, sd)
    rows, tot = [], {"words": 0, "raw_hits": 0, "extensions": 0, "bp_extended": 0, "hsps": 0}
    for strand, _, qq in H.strands(q):
        hsps, st = lzo.seed_hit_search(tab, qq, masked)
        rows += H.hsps_as_tsv_rows("query", strand, hsps)
        for k in tot:
            tot[k] += st[k]
    assert rows == gold
    for k in tot:
        assert tot[k] == gst[k], k


@pytest.mark.parametrize("case", ["synth200k", "synth_overlap", "adversarial"])
def test_gapped_stage_vs_reference_output(case):
    t, q = H.load_case(case)
    mine, cells = _oracle_gapped_blocks(t, [q])
    assert mine == H.lav_blocks(os.path.join(G, case + ".lav"))
    assert cells == H.load_stats(case)["dp_cells"]


@pytest.mark.skipif(lzo.ref_binary() is None, reason="oracle/_ref/lastz not built")
@pytest.mark.parametrize("seed,pattern,wt,step", [(21, H.DEFAULT_SEED, 1, 1), (22, "1111111111", 0, 1),
                                                   (23, "111101101111", 1, 3), (24, "11111111", 2, 1)])
def test_live_reference_hsps(tmp_path, seed, pattern, wt, step):
    """fresh inputs each parametrisation, straight against the reference binary"""
    t, q = seqio.synth_pair(50000, 70000, seed=seed, block_min=500, block_max=5000)
    tf, qf = str(tmp_path / "t.fa"), str(tmp_path / "q.fa")
    seqio.write_fasta(tf, [("target", t)]); seqio.write_fasta(qf, [("query", q)])
    trans = {0: "--notransition", 1: "--transition", 2: "--transition=2"}[wt]
    out = H.ref_run([tf, qf, "--nogapped", "--seed=" + pattern, trans, f"--step={step}",
                     "--format=general-:name2,start1,end1,start2,end2,strand2,score"])
    gold = [(f[0], int(f[1]), int(f[2]), int(f[3]), int(f[4]), f[5], int(f[6]))
            for f in (ln.split("\t") for ln in out.split("\n") if ln)]
    _, masked = H.scoring()
    sd = lzo.seed(pattern, wt)
    tab = lzo.Table(t, sd, step=step)
    rows = []
    for strand, _, qq in H.strands(q):
        hsps, _ = lzo.seed_hit_search(tab, qq, masked)
        rows += H.hsps_as_tsv_rows("query", strand, hsps)
    assert rows == gold


OPTION_CASES = {"noytrim": (["--noytrim"], dict(trim_to_peak=False)),
                "allgappedbounds": (["--allgappedbounds", "--gappedthresh=9000"], dict(all_bounds=True, score_thresh=9000)),
                "both": (["--noytrim", "--allgappedbounds", "--gappedthresh=6000"], dict(trim_to_peak=False, all_bounds=True, score_thresh=6000))}


def _option_pairs():
    """short pairs whose homology runs into the ends of the sequences, and the tandem-repeat obstacle course"""
    t, q = H.load_case("adversarial")
    t2, q2 = seqio.synth_pair(5000, 4200, seed=77, block_min=900, block_max=2500, homolog_frac=0.95)
    return {"adversarial_piece": (t[9000:14500], q[29500:34000]), "ends_q": (t2, q2[1200:2600]), "ends_t": (t2[300:3300], q2)}


def _oracle_blocks_with(t, q, **kw):
    sub, masked = H.scoring()
    tab = lzo.Table(t, lzo.seed())
    out = []
    for _, rev, qq in H.strands(q):
        hsps, _ = lzo.seed_hit_search(tab, qq, masked)
        al, ops, _ = lzo.gapped_extend(t, qq, sub, lzo.reduce_to_points(t, qq, sub, lzo.hsps_to_segments(hsps, rev)), **kw)
        if len(al):
            out.append((1, rev, H.blocks_of(al, ops)))
    return out


@pytest.mark.parametrize("opt", list(OPTION_CASES))
def test_untrimmed_ends_and_all_bounds_vs_reference_output(opt):
    """--noytrim / --allgappedbounds of the oracle against LAVs the pristine binary wrote (tests/golden/options_*.lav,
    made by tests/golden/make_golden.py --options) -- and, where oracle/_ref/lastz is present, against a fresh run"""
    flags, kw = OPTION_CASES[opt]
    changed = False
    for name, (t, q) in _option_pairs().items():
        mine = _oracle_blocks_with(t, q, **kw)
        assert mine == H.lav_blocks(os.path.join(G, f"options_{opt}_{name}.lav")), (opt, name)
        changed = changed or mine != _oracle_blocks_with(t, q, score_thresh=kw.get("score_thresh", 3000))
    assert changed                                            # the option decides something on these inputs


@pytest.mark.skipif(lzo.ref_binary() is None, reason="oracle/_ref/lastz not built")
@pytest.mark.parametrize("opt", list(OPTION_CASES))
def test_live_reference_untrimmed_ends_and_all_bounds(tmp_path, opt):
    flags, kw = OPTION_CASES[opt]
    for name, (t, q) in _option_pairs().items():
        tf, qf = str(tmp_path / "t.fa"), str(tmp_path / "q.fa")
        seqio.write_fasta(tf, [("target", t)]); seqio.write_fasta(qf, [("query", q)])
        lav = tmp_path / "o.lav"
        lav.write_text(H.ref_run([tf, qf] + flags))
        assert _oracle_blocks_with(t, q, **kw) == H.lav_blocks(str(lav)), (opt, name)
