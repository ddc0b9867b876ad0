"""N2: lzgpu_reduce_to_chain (lastz_amd/csrc/lz_chain_host.cpp) against what the pristine reference's reduce_to_chain
(src/chain.c:497) keeps -- tests/golden/chain_vectors.json, made by tests/golden/make_chain_vectors.py from runs of
oracle/_ref/lastz.  A host routine: these tests need no GPU (and prove the routine needs none).

The vectors discriminate: with the argument slip of src/chain.c:960-961 "corrected" 22 of the 88 cases differ, with `>=`
for `>` in the bucket scan 8, with buckets of four 15, with another pivot 36 (measured when the vectors were made).
"""
import json
import os

import numpy as np
import pytest

import helpers as H
from lastz_amd import lzgpu


@pytest.fixture(scope="module")
def lib():
    return lzgpu.Lib()


@pytest.fixture(scope="module")
def vectors():
    return json.load(open(os.path.join(H.GOLDEN, "chain_vectors.json")))


def segs(rows):
    a = np.zeros(len(rows), dtype=lzgpu.SEG_DTYPE)
    for i, (p1, p2, ln, s) in enumerate(rows):
        a[i] = (p1, p2, ln, s, 0)
    return a


def test_the_chain_the_reference_keeps(lib, vectors):
    assert len(vectors["cases"]) == 88
    for c in vectors["cases"]:
        rows = vectors["sets"][c["set"]]["anchors"]
        kept, _ = lib.reduce_to_chain(segs(rows), c["chain_diag"], c["chain_anti"], vectors["scale"], vectors["overlap_sub"])
        assert sorted(int(k) for k in kept) == c["kept"], (vectors["sets"][c["set"]]["pair"], c["chain_diag"], c["chain_anti"])


def test_order_of_the_input_does_not_matter_and_the_output_is_in_pos1_order(lib, vectors):
    rng = np.random.default_rng(3)
    for c in vectors["cases"][::7]:
        rows = vectors["sets"][c["set"]]["anchors"]
        a = segs(rows)
        perm = rng.permutation(len(a))
        kept, _ = lib.reduce_to_chain(a[perm], c["chain_diag"], c["chain_anti"], vectors["scale"], vectors["overlap_sub"])
        assert sorted(int(perm[k]) for k in kept) == c["kept"]
        p1 = a[perm][kept]["pos1"]
        assert (np.diff(p1.astype(np.int64)) > 0).all()        # a chain: strictly increasing in both sequences
        p2 = a[perm][kept]["pos2"]
        assert (np.diff(p2.astype(np.int64)) > 0).all()


def test_a_batch_of_problems_is_the_problems_one_by_one(lib, vectors):
    """lzgpu_reduce_to_chain_batch: every problem on a host thread of its own -- the same kept sets, the same scores"""
    by_pen = {}
    for c in vectors["cases"]:
        by_pen.setdefault((c["chain_diag"], c["chain_anti"]), []).append(c)
    for (d, a), cases in by_pen.items():
        sets = [segs(vectors["sets"][c["set"]]["anchors"]) for c in cases]
        got = lib.reduce_to_chain_batch(sets, d, a, vectors["scale"], vectors["overlap_sub"])
        assert len(got) == len(cases)
        for c, s, (kept, best) in zip(cases, sets, got):
            assert sorted(int(k) for k in kept) == c["kept"]
            one_kept, one_best = lib.reduce_to_chain(s, d, a, vectors["scale"], vectors["overlap_sub"])
            assert list(one_kept) == list(kept) and one_best == best
    assert lib.reduce_to_chain_batch([np.zeros(0, dtype=lzgpu.SEG_DTYPE), segs([(5, 9, 20, 3100)])])[1][1] == 3100


def test_score_of_the_chain(lib):
    """three anchors on one diagonal, no penalties: the chain is all of them and scores their sum; an overlap costs
    overlap_sub per overlapped base (src/lastz.c:3728-3733), which makes the overlapping one not worth taking"""
    a = segs([(100, 100, 50, 4000), (200, 200, 50, 3000), (300, 300, 50, 5000)])
    kept, best = lib.reduce_to_chain(a)
    assert list(kept) == [0, 1, 2] and best == 12000
    a = segs([(100, 100, 50, 4000), (120, 120, 50, 2000), (300, 300, 50, 5000)])      # 30 bases overlap: 30 * 91 > 2000
    kept, best = lib.reduce_to_chain(a)
    assert list(kept) == [0, 2] and best == 9000
    a = segs([(100, 100, 50, 4000), (140, 140, 50, 2000), (300, 300, 50, 5000)])      # 10 bases overlap: 2000 - 910 still pays
    kept, best = lib.reduce_to_chain(a)
    assert list(kept) == [0, 1, 2] and best == 11000 - 910


def test_edges(lib):
    kept, best = lib.reduce_to_chain(np.zeros(0, dtype=lzgpu.SEG_DTYPE))
    assert len(kept) == 0 and best == 0
    kept, best = lib.reduce_to_chain(segs([(5, 9, 20, 3100)]))
    assert list(kept) == [0] and best == 3100
    kept, best = lib.reduce_to_chain(segs([(5, 9, 20, 3100), (5, 9, 20, 3100), (5, 9, 20, 3100)]))   # identical records: one of them
    assert len(kept) == 1 and best == 3100
    kept, best = lib.reduce_to_chain(segs([(50, 10, 20, 3000), (10, 50, 20, 3000)]))      # crossing: never both
    assert len(kept) == 1
    kept, best = lib.reduce_to_chain(segs([(10, 10, 20, -5), (100, 100, 20, -7)]))        # nothing scores above zero: nothing kept (bestEnd stays noPred)
    assert len(kept) == 0 and best == 0


REF_BIN = os.path.join(H.ROOT, "oracle", "_ref", "lastz")


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/lastz not built")
def test_a_larger_pair_against_the_reference_binary(lib, tmp_path):
    """5 Mbp x 5 Mbp, ~8 k anchors per strand, three penalty settings: the rows `--chain` leaves are the rows the library keeps"""
    import subprocess
    from lastz_amd import seqio
    t, q = seqio.synth_pair(5_000_000, 5_000_000, seed=21)
    seqio.write_fasta(tmp_path / "t.fa", [("target", t)]); seqio.write_fasta(tmp_path / "q.fa", [("query", q)])
    fmt = "--format=general-:zstart1,end1,zstart2,end2,strand2,score"

    def rows(extra):
        out = subprocess.run([REF_BIN, "t.fa", "q.fa", "--nogapped", "--strand=plus", fmt] + extra, capture_output=True, text=True, cwd=tmp_path, check=True).stdout
        return [(int(f[0]), int(f[2]), int(f[1]) - int(f[0]), int(f[5])) for f in (ln.split("\t") for ln in out.splitlines())]
    anchors = rows([])
    assert len(anchors) > 3000
    for d, a in ((0, 0), (20, 20), (400, 3)):
        kept, _ = lib.reduce_to_chain(segs(anchors), d, a)
        assert sorted(anchors[k] for k in kept) == sorted(rows(["--chain=%d,%d" % (d, a)]))
