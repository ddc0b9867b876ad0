"""lastz_amd/lav.py: the LAV text of a set of alignments as the reference writes it (src/lav.c:57-127, 235-300) -- what lets bench.py and
the GPU suite fingerprint the alignments lzgpu_gapped_extend_batch RETURNS against the file the pristine reference wrote (VERDICT r4 #3d)."""
import os

import numpy as np
import pytest

import helpers as H
from oracle import lzo
from lastz_amd import lav, lzgpu


@pytest.mark.parametrize("case", ["synth200k", "adversarial"])
def test_rendered_lav_is_the_references_file(case):
    t, q = H.load_case(case)
    sub, masked = H.scoring()
    tab = lzo.Table(t, lzo.seed())
    per, qs = [], []
    for _, rev, qq in H.strands(q):
        hsps, _ = lzo.seed_hit_search(tab, qq, masked)
        segs = lzo.hsps_to_segments(hsps, rev)
        al, ops, _ = lzo.gapped_extend(t, qq, sub, lzo.reduce_to_points(t, qq, sub, segs), ydrop=9400)
        per.append((al, ops)); qs.append(qq)
    gold = open(os.path.join(H.GOLDEN, case + ".lav")).read()
    mine = lav.render(t, qs, per, sub)
    a, b = mine.split("\n"), gold.split("\n")
    assert a[:2] == b[:2] and a[3:] == b[3:]                  # everything but the command line
    assert lav.fingerprint(mine) == lav.fingerprint(gold)
    assert lav.compare(per, gold)["equal"]
    # a changed score, a shifted piece, a dropped block: each is seen
    al2 = per[0][0].copy(); al2["s"][0] += 1
    assert not lav.compare([(al2, per[0][1]), per[1]], gold)["equal"]
    assert lav.fingerprint(lav.render(t, qs, [(al2, per[0][1]), per[1]], sub)) != lav.fingerprint(gold)
    assert not lav.compare([(per[0][0][1:], per[0][1]), per[1]], gold)["equal"]


def test_an_insert_behind_a_delete_leaves_an_empty_piece():
    """src/lav.c:268-283: since 1.03.55 the empty run between a delete and an insert is printed (l b1 b2 b1-1 b2-1 0)"""
    t = np.frombuffer(b"ACGTACGTACGTACGTACGTACGT", dtype=np.uint8)
    q = np.frombuffer(b"ACGTACGTACTTTGTACGTACGT", dtype=np.uint8)
    al = np.zeros(1, dtype=lzgpu.ALIGN_DTYPE)
    ops = np.array([3 | (10 << 2), 2 | (2 << 2), 1 | (3 << 2), 3 | (5 << 2)], dtype=np.uint32)       # 10 subs, delete 2, insert 3, 5 subs
    al[0] = (1, 1, 17, 18, 1234, 4, 0)
    text = lav.render(t, [q, q], [(al, ops), (al[:0], ops[:0])], H.scoring()[0])
    ls = [ln.split() for ln in text.split("\n") if ln.startswith("  l ")]
    assert [x[1:5] for x in ls] == [["1", "1", "10", "10"], ["13", "11", "12", "10"], ["13", "14", "17", "18"]]
    assert ls[1][5] == "0" and ls[0][5] == "100"
    st = lav.parse(text)
    assert len(st) == 2 and st[0]["rev2"] == 0 and st[1]["rev2"] == 1 and len(st[0]["blocks"]) == 1 and st[0]["blocks"][0]["score"] == 1234
