"""Host logic of the one-process-per-GPU launcher (lastz_amd/multi.py) on CPU: the reference's golden LAV files cut into
per-rank outputs and merged back, the unit plan, FASTA lengths."""
import os
import numpy as np

from lastz_amd import multi, shard, seqio
import helpers as H


def _as_rank_output(head, units):
    return "#:lav\n" + head[0] + "".join("#:lav\n" + u for _, u in units) + head[1] + "#:eof\n"


def test_merge_restores_the_reference_order():
    for name in ("base_test.default.lav", "base_test.chained.lav", "base_test.interpolated.lav"):
        text = open(os.path.join(H.GOLDEN, name)).read()
        head, units = multi.split_lav(text)
        assert [k for k, _ in units] == sorted(k for k, _ in units)          # queries in file order, + before -
        for world in (2, 3):
            parts = [[u for i, u in enumerate(units) if i % world == r] for r in range(world)]
            outs = [_as_rank_output(head, list(reversed(p))) for p in parts]   # a rank's own order does not matter
            assert multi.merge_lav(outs[::-1]) == text
        assert multi.merge_lav([_as_rank_output(head, []), text]) == text    # a rank without alignments
        if "\nm {\n" in text:
            assert head[1].startswith("m {\n")
            continue
        withm = text.replace("#:eof\n", "m {\n  n 0\n}\n#:eof\n")            # the m-stanza that closes a target (src/lastz.c:1761)
        hm, um = multi.split_lav(withm)
        assert hm[1] == "m {\n  n 0\n}\n" and [u for _, u in um] == [u for _, u in units]
        assert multi.merge_lav([_as_rank_output(hm, um[1::2]), _as_rank_output(hm, um[::2])]) == withm


def test_fasta_lengths_and_plan(tmp_path):
    rng = np.random.default_rng(1)
    seqs = [("s%d" % i, np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)]) for i, n in enumerate((1000, 61, 60, 5, 12345))]
    seqio.write_fasta(tmp_path / "q.fa", seqs)
    lens = multi.fasta_lengths(str(tmp_path / "q.fa"))
    assert lens == [1000, 61, 60, 5, 12345]
    plan = shard.plan_units(lens, 2)
    assert sorted(u for p in plan for u in p) == [(i, s) for i in range(5) for s in (0, 1)]
    loads = [sum(lens[i] for i, _ in p) for p in plan]
    assert abs(loads[0] - loads[1]) <= max(lens)
