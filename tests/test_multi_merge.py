"""Host logic of the one-process-per-GPU launcher (lastz_amd/multi.py) on CPU: the reference's golden LAV files cut into
per-rank outputs and merged back, the unit plan, FASTA lengths."""
import os
import numpy as np
import pytest

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


REF_BIN = os.path.join(H.ROOT, "oracle", "_ref", "lastz")


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/lastz not built")
def test_rank_query_files_keep_contig_numbers_and_output(tmp_path):
    """The launcher's per-rank query files (records of other ranks reduced to their header lines) through the
    PRISTINE reference binary: whole sequences per rank, so that each process prints exactly its units, and the
    merged LAV is the single-process LAV byte for byte (contig numbers, file name in the s-stanzas included)."""
    import subprocess
    t = os.path.join(H.GOLDEN, "pseudocat.fa"); q = os.path.join(H.GOLDEN, "pseudopig.fa")
    for args in ([], ["--chain"], ["--nogapped", "--strand=plus"]):
        single = subprocess.run([REF_BIN, t, q] + args, capture_output=True, text=True, check=True).stdout
        for ranks in (2, 3):
            merged, errs, plan = multi.run(t, q, args, ranks=ranks, lastz=REF_BIN, whole_sequences=True)
            assert merged == single
            assert multi.run.last["split"] and sum(multi.run.last["owned_bases"]) == sum(multi.fasta_lengths(q))
            assert all("contains an empty sequence" in e for e in errs)          # what a rank skips, it skips unparsed


def test_index_of_a_fasta_and_rank_files(tmp_path):
    rng = np.random.default_rng(5)
    seqs = [("s%d extra words" % i, np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)]) for i, n in enumerate((300, 61, 1, 777))]
    seqio.write_fasta(tmp_path / "q.fa", seqs)
    src = str(tmp_path / "q.fa")
    idx = multi.fasta_index(src)
    assert [r[3] for r in idx] == [300, 61, 1, 777]
    multi.write_rank_query(src, idx, {1, 3}, str(tmp_path / "r.fa"))
    text = open(tmp_path / "r.fa").read()
    assert text.count(">") == 4 and [len("".join(blk.split("\n")[1:])) for blk in text.split(">")[1:]] == [0, 61, 0, 777]
    assert multi.split_spec("a/b.fa[unmask][multi]") == ("a/b.fa", "[unmask][multi]") and multi.split_spec("x.fa") == ("x.fa", "")
    with pytest.raises(ValueError):
        multi.check_supported(src, ["--format=maf"])
    with pytest.raises(ValueError):
        multi.check_supported(src, [])                                           # several target sequences, no [multi]
    multi.check_supported(src + "[multi]", ["--format=lav"])


def test_line_oriented_formats_merge_by_unit_markers():
    """MAF / AXT / general: the bound binary prints '#lzgpu-unit <contig> <strand>' before a unit's records
    (integration/lzgpu_shim.c::unit_marker); the launcher sorts the units and counts AXT's numbers again"""
    head = "# lastz /tmp/x/query.rank0.fa --format=axt\n#\n"
    rec = lambda n, q, s: "%d t 1 5 %s 1 5 %s 400\nACGTA\nACGTA\n\n" % (n, q, s)
    r0 = head + "#lzgpu-unit 1 1\n" + rec(0, "qa", "-") + rec(1, "qa", "-") + "#lzgpu-unit 3 0\n" + rec(2, "qc", "+")
    r1 = head.replace("rank0", "rank1") + "#lzgpu-unit 1 0\n" + rec(0, "qa", "+") + "#lzgpu-unit 2 0\n#lzgpu-unit 2 1\n" + rec(1, "qb", "-")
    merged = multi.merge_marked([r0, r1], "axt", [("/tmp/x/query.rank0.fa", "q.fa"), ("/tmp/x/query.rank1.fa", "q.fa")])
    want = ("# lastz q.fa --format=axt\n#\n" + rec(0, "qa", "+") + rec(1, "qa", "-") + rec(2, "qa", "-")
            + rec(3, "qb", "-") + rec(4, "qc", "+"))
    assert merged == want
    # MAF keeps its records as they are
    m0 = "##maf version=1\n#lzgpu-unit 2 0\na score=9\ns t 0 5 + 9 ACGTA\ns qb 0 5 + 7 ACGTA\n\n"
    m1 = "##maf version=1\n#lzgpu-unit 1 0\na score=7\ns t 0 5 + 9 ACGTA\ns qa 0 5 + 7 ACGTA\n\n"
    assert multi.merge_marked([m0, m1], "maf") == "##maf version=1\n" + m1.split("\n", 2)[2] + m0.split("\n", 2)[2]
    # a rank that produced nothing at all contributes its header only
    assert multi.merge_marked(["##maf version=1\n", m1], "maf") == "##maf version=1\n" + m1.split("\n", 2)[2]
    # SAM: every rank prints the target's @SQ lines once, behind its first unit marker; the job has them once, behind @HD
    s0 = "@HD\tVN:1.0\tSO:unsorted\n#lzgpu-unit 2 0\n@SQ\tSN:t\tLN:9\nqb\t0\tt\t1\t255\t5M\t*\t0\t0\tACGTA\t*\n"
    s1 = "@HD\tVN:1.0\tSO:unsorted\n#lzgpu-unit 1 0\n@SQ\tSN:t\tLN:9\nqa\t0\tt\t2\t255\t5M\t*\t0\t0\tCGTAC\t*\n#lzgpu-unit 1 1\nqa\t16\tt\t3\t255\t5M\t*\t0\t0\tGTACG\t*\n"
    assert multi.merge_marked([s0, s1], "sam") == ("@HD\tVN:1.0\tSO:unsorted\n@SQ\tSN:t\tLN:9\nqa\t0\tt\t2\t255\t5M\t*\t0\t0\tCGTAC\t*\n"
                                                    "qa\t16\tt\t3\t255\t5M\t*\t0\t0\tGTACG\t*\nqb\t0\tt\t1\t255\t5M\t*\t0\t0\tACGTA\t*\n")


def test_formats_the_launcher_takes_and_refuses(tmp_path):
    t = tmp_path / "t.fa"; t.write_text(">t\nACGT\n")
    for ok in (["--format=maf"], ["--format=MAF-"], ["--axt"], ["--format=general:name1,start1,name2"], ["--format=general-"],
               ["--format=cigar"], ["--format=differences"], ["--format=sam"], ["--format=softsam-"], []):
        multi.check_supported(str(t), ok)
        assert (multi.output_format(ok) == "lav") == (ok == [])
    for bad in (["--format=rdotplot"], ["--format=text"], ["--format=lav+text"], ["--format=blastn"], ["--format=gfa"]):
        with pytest.raises(ValueError):
            multi.check_supported(str(t), bad)


def test_quantum_is_read_off_the_sequence_specifier_not_off_a_substring(tmp_path):
    """a query NAMED quantum_reads.fa is an ordinary query; [quantum] actions and .qdna files are refused"""
    t = tmp_path / "t.fa"; t.write_text(">t\nACGT\n")
    q = tmp_path / "quantum_reads.fa"; q.write_text(">q\nACGT\n")
    multi.check_supported(str(t), ["--format=maf", "--output-note=quantum"], str(q))
    multi.check_supported(str(t), [], str(q) + "[unmask]")
    for bad in (str(q) + "[quantum]", str(q) + "[unmask][quantum=x.codes]", str(q) + "[unmask,quantum]", str(tmp_path / "reads.qdna")):
        with pytest.raises(ValueError):
            multi.check_supported(str(t), [], bad)
    for bad in (["--anyornone"], ["--segments=x"], ["--chores=y"]):
        with pytest.raises(ValueError):
            multi.check_supported(str(t), bad, str(q))
