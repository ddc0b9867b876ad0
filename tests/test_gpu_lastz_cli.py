"""The drop-in claim, end to end: the REFERENCE's own command lines (src/Makefile:208-217,295-400)
run through integration/_build/lastz_gpu -- the reference's host code with its three hot-path entry
points bound to liblzgpu.so by integration/lzgpu_shim.c -- must give the reference's golden LAV
byte for byte (modulo the first line of the d-stanza, which holds the command line; that is what
tools/lav_compare.py ignores too), and must agree byte for byte with the pristine binary on
MAF / AXT / general output of a fresh synthetic pair.  Needs an MI355X and the prebuilt binaries."""
import os
import shutil
import subprocess
import pytest

from lastz_amd import seqio
import helpers as H
from lavparse import normalize_lav

pytestmark = pytest.mark.gpu
GPU_BIN = os.path.join(H.ROOT, "integration", "_build", "lastz_gpu")
REF_BIN = os.path.join(H.ROOT, "oracle", "_ref", "lastz")
needs_bins = pytest.mark.skipif(not (os.path.exists(GPU_BIN) and os.path.exists(REF_BIN)),
                                reason="integration/_build/lastz_gpu not built (needs /root/reference at build time)")


@pytest.fixture(scope="module")
def sandbox(tmp_path_factory):
    d = tmp_path_factory.mktemp("lz")
    os.makedirs(d / "test_data"); os.makedirs(d / "src")
    for f in ("pseudocat.fa", "pseudopig.fa", "aglobin.2bit"):
        shutil.copy(os.path.join(H.GOLDEN, f), d / "test_data" / f)
    return d


def run(binary, args, cwd, env_extra=None):
    env = dict(os.environ); env.update(env_extra or {})
    p = subprocess.run([binary] + args, cwd=cwd, capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    return p.stdout, p.stderr


CASES = [("base_test.default.lav", [], {"table": 1, "search": 6, "gapped": 6}),
         ("base_test.hsp.lav", ["C=3", "W=8", "T=0"], {"table": 1, "search": 6}),
         ("base_test.extended.lav", ["C=2", "W=8", "T=0"], {"table": 1, "search": 6}),
         ("base_test.chained.lav", ["C=1", "W=8", "T=0"], {"table": 1, "search": 6, "chain": 6}),
         ("base_test.interpolated.lav", ["C=2", "W=8", "T=0", "H=2200"], {"table": 1, "search": 6}),     # tweener second pass
         ("base_test.hits.lav", ["W=8", "T=0", "--plus", "--nogfextend", "--nogapped"], {"table": 1})]


@needs_bins
@pytest.mark.parametrize("golden,flags,on_gpu", CASES, ids=[c[0] for c in CASES])
def test_reference_command_lines(sandbox, golden, flags, on_gpu):
    out, err = run(GPU_BIN, ["../test_data/pseudocat.fa", "../test_data/pseudopig.fa"] + flags, sandbox / "src",
                   {"LZGPU_VERBOSE": "1"})
    want = open(os.path.join(H.GOLDEN, golden)).read()
    assert normalize_lav(out) == normalize_lav(want)
    for stage, n in on_gpu.items():                 # the GPU path really ran (no quiet reference fallback)
        assert err.count(f"[lzgpu] {stage}: ") >= n
        assert err.count(f"[lzgpu] {stage}: done on the GPU") + err.count(f"[lzgpu] {stage}: built on the GPU") + err.count(f"[lzgpu] {stage}: done by the library") >= n, err[-1500:]


@needs_bins
@pytest.mark.parametrize("fmt", ["lav", "maf", "axt", "general"])
def test_same_bytes_as_pristine_binary(sandbox, fmt):
    t, q = seqio.synth_pair(1_500_000, 1_200_000, seed=41)
    seqio.write_fasta(sandbox / "t.fa", [("target", t)])
    seqio.write_fasta(sandbox / "q.fa", [("q1", q[:700000]), ("q2", q[700000:])])
    args = ["t.fa", "q.fa", "--format=" + fmt, "--ydrop=9430"]
    a, err = run(GPU_BIN, args, sandbox, {"LZGPU_VERBOSE": "1"})
    b, _ = run(REF_BIN, args, sandbox)
    assert "[lzgpu] gapped: done on the GPU" in err and "[lzgpu] search: done on the GPU" in err
    strip = (lambda s: normalize_lav(s)) if fmt == "lav" else (lambda s: "\n".join(l for l in s.split("\n") if not l.startswith("#")))
    assert strip(a) == strip(b)
    assert len(a) > 2000


@needs_bins
@pytest.mark.parametrize("chain", ["--chain", "--chain=20,20", "--chain=400,3", "--chain=3,400"])
def test_chaining_by_the_library(sandbox, chain):
    """N2: try_reduce_to_chain of the bound binary is the library's host routine (lastz_amd/csrc/lz_chain_host.cpp); HSPs of the GPU
    search in, chain out, gapped extension of the chain on the GPU: the pristine binary's bytes"""
    t, q = seqio.synth_pair(2_000_000, 2_000_000, seed=43)
    q[300_000:420_000] = q[900_000:1_020_000]                   # a duplication: competing off-diagonal HSPs
    seqio.write_fasta(sandbox / "tc.fa", [("target", t)]); seqio.write_fasta(sandbox / "qc.fa", [("query", q)])
    for tail in (["--nogapped", "--format=general-:start1,end1,start2,end2,strand2,score"], ["--ydrop=9430", "--format=axt"]):
        args = ["tc.fa", "qc.fa", chain] + tail
        a, err = run(GPU_BIN, args, sandbox, {"LZGPU_VERBOSE": "1"})
        b, _ = run(REF_BIN, args, sandbox)
        assert err.count("[lzgpu] chain: done by the library") == 2 and "[lzgpu] chain: reference path" not in err, err[-1500:]
        strip = lambda s: "\n".join(l for l in s.split("\n") if not l.startswith("#"))
        assert strip(a) == strip(b) and len(a) > 2000


@needs_bins
def test_wide_bands_stay_on_the_gpu(sandbox):
    """O=200 E=5 needs bands of > 2048 columns: the HBM-ring DP runs them (round 1 declined the whole stage)"""
    args = ["../test_data/pseudocat.fa", "../test_data/pseudopig.fa", "O=200", "E=5", "--format=axt"]
    a, err = run(GPU_BIN, args, sandbox / "src", {"LZGPU_VERBOSE": "1"})
    b, _ = run(REF_BIN, args, sandbox / "src")
    assert err.count("[lzgpu] gapped: done on the GPU") == 6 and "declined" not in err, err[-1500:]
    strip = lambda s: "\n".join(l for l in s.split("\n") if not l.startswith("#"))
    assert strip(a) == strip(b) and len(a) > 2000


@needs_bins
@pytest.mark.parametrize("extra", [[], ["--notrivial"], ["--format=maf", "--strand=plus"]], ids=["plain", "notrivial", "maf-plus"])
def test_identical_sequences(sandbox, extra):
    """a sequence against itself WITHOUT --self, and against a soft-masked copy: the trivial self-alignment path
    (src/gapped_extend.c:1152-1189) runs on the GPU and gives the pristine binary's bytes"""
    t, _ = seqio.synth_pair(300_000, 1000, seed=51)
    t[100_000:130_000] = t[20_000:50_000]                       # a repeat: off-diagonal alignments bounded by the trivial one
    seqio.write_fasta(sandbox / "ti.fa", [("same", t)])
    m = t.copy(); m[5000:9000] |= 0x20
    seqio.write_fasta(sandbox / "tim.fa", [("same", m)])
    for q in ("ti.fa", "tim.fa"):
        args = ["ti.fa", q, "--ydrop=9430"] + extra
        a, err = run(GPU_BIN, args, sandbox, {"LZGPU_VERBOSE": "1"})
        b, _ = run(REF_BIN, args, sandbox)
        assert "[lzgpu] gapped: done on the GPU" in err and "declined" not in err, err[-1500:]
        if "--format=maf" in extra:
            strip = lambda s: "\n".join(l for l in s.split("\n") if not l.startswith("#"))
            assert strip(a) == strip(b)
        else:
            assert normalize_lav(a) == normalize_lav(b)
        assert len(a) > 300


@needs_bins
def test_table_cache_directory(sandbox, tmp_path):
    """LZGPU_TABLE_CACHE: the first run writes the target's table file, the second loads it; same bytes out."""
    t, q = seqio.synth_pair(400_000, 300_000, seed=43)
    seqio.write_fasta(sandbox / "tc.fa", [("target", t)]); seqio.write_fasta(sandbox / "qc.fa", [("q", q)])
    cache = tmp_path / "cache"; os.makedirs(cache)
    env = {"LZGPU_VERBOSE": "1", "LZGPU_TABLE_CACHE": str(cache)}
    a, ea = run(GPU_BIN, ["tc.fa", "qc.fa", "--format=maf"], sandbox, env)
    assert "[lzgpu] table: built on the GPU" in ea and len(os.listdir(cache)) == 1
    b, eb = run(GPU_BIN, ["tc.fa", "qc.fa", "--format=maf"], sandbox, env)
    assert "[lzgpu] table: loaded from the table cache" in eb and len(os.listdir(cache)) == 1
    c, ec = run(GPU_BIN, ["tc.fa", "qc.fa", "--format=maf", "--step=2"], sandbox, env)     # another table: another file
    assert "loaded from the table cache" not in ec and len(os.listdir(cache)) == 2
    ref, _ = run(REF_BIN, ["tc.fa", "qc.fa", "--format=maf"], sandbox)
    strip = lambda s: "\n".join(l for l in s.split("\n") if not l.startswith("#"))
    assert strip(a) == strip(b) == strip(ref) and len(a) > 1000


@needs_bins
@pytest.mark.parametrize("qmulti", [False, True], ids=["target-multi", "both-multi"])
def test_partitioned_sequences_gapped_on_gpu(sandbox, qmulti):
    """file[multi] (whole assemblies as one target): the gapped stage keeps every extension inside the
    partition holding its anchor (src/gapped_extend.c:1356-1372); same bytes as the pristine binary."""
    t, q = seqio.synth_pair(900_000, 800_000, seed=47)
    seqio.write_fasta(sandbox / "tm.fa", [("c1", t[:250_000]), ("c2", t[250_000:610_000]), ("c3", t[610_000:])])
    seqio.write_fasta(sandbox / "qm.fa", [("q1", q[:300_000]), ("q2", q[300_000:])])
    args = ["tm.fa[multi]", "qm.fa" + ("[multi]" if qmulti else ""), "--format=maf", "--ydrop=9430"]
    a, err = run(GPU_BIN, args, sandbox, {"LZGPU_VERBOSE": "1"})
    b, _ = run(REF_BIN, args, sandbox)
    assert "[lzgpu] gapped: done on the GPU" in err and "[lzgpu] search: done on the GPU" in err, err[-1500:]
    strip = lambda s: "\n".join(l for l in s.split("\n") if not l.startswith("#"))
    assert strip(a) == strip(b) and len(a) > 5000


@needs_bins
def test_gapped_stage_alone_from_saved_segments(sandbox):
    """src/Makefile:384-400: HSPs saved with --format=segments, gapped stage run from --segments=<file>
    (B3 in isolation) must give base_test.default.lav"""
    src = sandbox / "src"
    hsps, _ = run(GPU_BIN, ["../test_data/pseudocat.fa", "../test_data/pseudopig.fa", "--nogapped", "--format=segments"], src)
    open(src / "hsps.segments", "w").write(hsps)
    out, err = run(GPU_BIN, ["../test_data/pseudocat.fa", "../test_data/pseudopig.fa", "--segments=hsps.segments"], src,
                   {"LZGPU_VERBOSE": "1"})
    want = open(os.path.join(H.GOLDEN, "base_test.default.lav")).read()
    drop = lambda s: "\n".join(l for l in normalize_lav(s).split("\n"))
    assert drop(out) == drop(want)
    assert err.count("[lzgpu] gapped: done on the GPU") >= 6


@needs_bins
@pytest.mark.parametrize("args", [["../test_data/aglobin.2bit/human", "../test_data/aglobin.2bit/cow"],
                                  ["../test_data/aglobin.2bit/human[20000..60000]", "../test_data/aglobin.2bit/cow", "--step=3", "--seed=match12"],
                                  ["../test_data/aglobin.2bit/human", "../test_data/aglobin.2bit/cow", "--notransition", "--nogapped", "--format=maf"],
                                  ["../test_data/aglobin.2bit/cow", "../test_data/aglobin.2bit/human", "--hspthresh=2200", "--ydrop=5000", "--gappedthresh=4000", "--format=axt"],
                                  ["../test_data/aglobin.2bit/human", "--self", "--chain"]],
                         ids=["human-cow", "subrange-step3-match12", "notransition-maf", "thresholds-axt", "self-chain"])
def test_real_dna_soft_masked_2bit(sandbox, args):
    """aglobin.2bit (the reference's real-DNA fixture: lower-case = soft-masked): the GPU-bound binary and
    the pristine binary give the same bytes for a spread of option sets; self-alignment (BASELINE.json
    config 1) is outside the fast path and must still be right (reference routines via the shim)."""
    src = sandbox / "src"
    a, err = run(GPU_BIN, args, src, {"LZGPU_VERBOSE": "1"})
    b, _ = run(REF_BIN, args, src)
    na = normalize_lav(a) if "--format" not in " ".join(args) else a
    nb = normalize_lav(b) if "--format" not in " ".join(args) else b
    strip = lambda s: "\n".join(l for l in s.split("\n") if not l.startswith("#"))
    assert strip(na) == strip(nb)
    assert len(a) > 500
    if "--self" not in args:
        assert "[lzgpu] table: built on the GPU" in err and "[lzgpu] search: done on the GPU" in err


@needs_bins
def test_bench_pair_lav_fingerprint(tmp_path):
    """BASELINE.json configs[2] at full size: the LAV that lastz_gpu writes for the 50 Mbp x 50 Mbp bench pair has the
    SHA-256 of the pristine reference's (tests/golden/bench50m.sha.json; the reference needs 40 minutes for it)"""
    import json, bench
    gold = json.load(open(os.path.join(H.GOLDEN, "bench50m.sha.json")))
    t, q = seqio.synth_pair(gold["tlen"], gold["qlen"], seed=gold["seed"])
    seqio.write_fasta(tmp_path / "t.fa", [("target", t)]); seqio.write_fasta(tmp_path / "q.fa", [("query", q)])
    out, err = run(GPU_BIN, ["t.fa", "q.fa", "--ydrop=9430"], tmp_path, {"LZGPU_VERBOSE": "1"})
    assert err.count("[lzgpu] gapped: done on the GPU") == 2 and err.count("[lzgpu] search: done on the GPU") == 2
    assert out.count("\na {") == gold["lav_blocks"]
    assert bench.lav_fingerprint(out) == gold["lav_sha"]


@needs_bins
@pytest.mark.parametrize("flags,notes", [(["--noytrim"], {"done on the GPU": 1}),
                                         (["--allgappedbounds", "--gappedthresh=9000"], {"done on the GPU": 1}),
                                         (["--querydepth=keep,nowarn:1.5"], {"done on the GPU": 1}),
                                         (["--querydepth=keep,nowarn:0.02"], {"declined, reference path": 1})],
                         ids=["noytrim", "allgappedbounds", "querydepth-not-reached", "querydepth-reached"])
def test_gapped_options_that_used_to_be_declined(sandbox, flags, notes):
    """round 3: untrimmed ends, all alignments as bounds and a limit on the paired bases run on the device (the limit,
    once exceeded, hands the stage back to the reference's routine, which warns and truncates as --querydepth asks);
    byte-identical to the pristine binary"""
    t, q = seqio.synth_pair(300_000, 250_000, seed=43, block_min=400, block_max=6000)
    seqio.write_fasta(sandbox / "to.fa", [("target", t)]); seqio.write_fasta(sandbox / "qo.fa", [("query", q)])
    args = ["to.fa", "qo.fa", "--ydrop=9430", "--format=maf"] + flags
    out, err = run(GPU_BIN, args, sandbox, {"LZGPU_VERBOSE": "1"})
    ref, _ = run(REF_BIN, args, sandbox)
    drop = lambda s: "".join(l for l in s.splitlines(True) if not l.startswith("#"))
    assert drop(out) == drop(ref) and out.count("\na score=") > 3
    for how, n in notes.items():
        assert err.count("[lzgpu] gapped: " + how) >= n, err[-1500:]
    if "declined, reference path" not in notes:
        assert "[lzgpu] gapped: declined" not in err and "[lzgpu] gapped: reference path" not in err


PART_CASES = {
    # seq1 partitioned, seq2 = one of its partitions (identical_partition_of_sequence): one trivial alignment in front
    "partition-of": (["pm.fa[multi]", "pq2.fa"], {"done on the GPU": 1}),
    "partition-of-notrivial": (["pm.fa[multi]", "pq2.fa", "--notrivial"], {"done on the GPU": 1}),
    # both partitioned, every pair (k, k) identical (identical_partitioned_sequences): one trivial alignment per pair
    "all-pairs": (["pm.fa[multi]", "pm.fa[multi]"], {"done on the GPU": 1}),
    "all-pairs-notrivial": (["pm.fa[multi]", "pm.fa[multi]", "--notrivial"], {"done on the GPU": 1}),
    # both partitioned, the same sequences in another order: no partitioned triviality; without --notrivial nothing special
    "reordered": (["pm.fa[multi]", "pr.fa[multi]"], {"done on the GPU": 1}),
    # ... and with it the reference tells trivial alignments by sequence NAME at output time: a result that holds a
    # candidate goes back to the reference's routine
    "reordered-notrivial": (["pm.fa[multi]", "pr.fa[multi]", "--notrivial"], {"declined, reference path": 1}),
}


@needs_bins
@pytest.mark.parametrize("case", list(PART_CASES))
def test_identical_partitions(sandbox, case):
    """the trivial alignments of partitioned sequences (src/gapped_extend.c:1118-1290, :1483-1545) on the device path
    (round 2 declined any pair holding two identical partitions); byte-identical to the pristine binary"""
    t, q = seqio.synth_pair(500_000, 400_000, seed=53)
    parts = [("p1", t[:150_000]), ("p2", t[150_000:330_000]), ("p3", t[330_000:])]
    seqio.write_fasta(sandbox / "pm.fa", parts)
    seqio.write_fasta(sandbox / "pq2.fa", [("p2", parts[1][1])])
    seqio.write_fasta(sandbox / "pr.fa", [parts[2], parts[0], parts[1]])
    args, notes = PART_CASES[case]
    args = args + ["--format=maf", "--ydrop=9430"]
    a, err = run(GPU_BIN, args, sandbox, {"LZGPU_VERBOSE": "1"})
    b, _ = run(REF_BIN, args, sandbox)
    strip = lambda s: "\n".join(l for l in s.split("\n") if not l.startswith("#"))
    assert strip(a) == strip(b)
    for how, n in notes.items():
        assert err.count("[lzgpu] gapped: " + how) >= n, err[-1500:]
    if "declined, reference path" not in notes:
        assert "[lzgpu] gapped: declined" not in err
