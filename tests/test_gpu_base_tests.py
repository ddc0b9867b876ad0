"""The reference's whole `make base_tests` list (src/Makefile:295-591), command line by command line, through
integration/_build/lastz_gpu -- the reference's host code bound to liblzgpu.so -- and through the pristine binary on
the same box: the outputs must be identical byte for byte (whatever the format: LAV, AXT, MAF, GFA, segments), the
reference's own golden files must match where the Makefile compares them with plain `diff` or lav_compare, and for
every case the test pins WHICH stages ran on the GPU and which were declined to the reference's routine (and so
that a fall-back never goes unnoticed, nor a stage that should decline runs on the device).

`base_test_float` is the one target left out: it needs lastz_D (floating-point scores, another build of the
reference); the library computes in the reference's default s32 arithmetic only.

Inputs and goldens are data fixtures copied from the reference's test_data/ (tests/golden/)."""
import os
import shutil
import subprocess
import pytest

import helpers as H
from lavparse import normalize_lav

pytestmark = pytest.mark.gpu
GPU_BIN = os.path.join(H.ROOT, "integration", "_build", "lastz_gpu")
REF_BIN = os.path.join(H.ROOT, "oracle", "_ref", "lastz")
needs_bins = pytest.mark.skipif(not (os.path.exists(GPU_BIN) and os.path.exists(REF_BIN)),
                                reason="integration/_build/lastz_gpu / oracle/_ref/lastz not built (need /root/reference at build time)")

DATA = ["pseudocat.fa", "pseudopig.fa", "aglobin.2bit", "pseudopig.2bit", "fake_apple.fa", "fake_orange_reads.fa",
        "shorties.2bit", "shorties.fa", "shorties.names", "base_test.anchors.anchors", "base_test.anchors_multi.anchors",
        "pseudopig.n.mask"]


@pytest.fixture(scope="module")
def sandbox(tmp_path_factory):
    d = tmp_path_factory.mktemp("base_tests")
    os.makedirs(d / "test_data"); os.makedirs(d / "src"); os.makedirs(d / "test_results")
    for f in DATA:
        shutil.copy(os.path.join(H.GOLDEN, f), d / "test_data" / f)
    return d


def run(binary, args, cwd, stdin=None, verbose=False):
    env = dict(os.environ)
    if verbose:
        env["LZGPU_VERBOSE"] = "1"
    p = subprocess.run([binary] + args, cwd=cwd, capture_output=True, text=True, env=env, timeout=600, input=stdin)
    assert p.returncode == 0, p.stderr[-2000:]
    return p.stdout, p.stderr


def notes(err):
    """{stage: {how: count}} from the shim's LZGPU_VERBOSE lines"""
    out = {}
    for line in err.split("\n"):
        if line.startswith("[lzgpu] "):
            stage, how = line[8:].split(": ", 1)
            out.setdefault(stage, {}).setdefault(how, 0)
            out[stage][how] += 1
    return out


T = "../test_data/"
# name, args, golden file (None: compared with the pristine binary only), how the golden is compared, stdin file,
# expected stage notes {stage: {how: count}} -- "done on the GPU" / "built on the GPU" vs "reference path"
CASES = [
    ("hits", [T + "pseudocat.fa", T + "pseudopig.fa", "W=8", "T=0", "--plus", "--nogfextend", "--nogapped"], "base_test.hits.lav", "lav", None,
     {"table": {"built on the GPU": 1, "copied to the host for a reference routine": 1}, "search": {"reference path": 3}}),   # --nogfextend: another hit processor
    ("hsp", [T + "pseudocat.fa", T + "pseudopig.fa", "C=3", "W=8", "T=0"], "base_test.hsp.lav", "lav", None,
     {"table": {"built on the GPU": 1}, "search": {"done on the GPU": 6}}),
    ("adaptive_k", [T + "aglobin.2bit/human", T + "aglobin.2bit/cow", "C=3", "W=8", "T=0", "--noentropy", "K=top50%", "--gfa"], None, None, None,
     {"table": {"built on the GPU": 1, "copied to the host for a reference routine": 1}, "search": {"reference path": 2}}),   # K=top50%: adaptive threshold
    ("default", [T + "pseudocat.fa", T + "pseudopig.fa"], "base_test.default.lav", "lav", None,
     {"table": {"built on the GPU": 1}, "search": {"done on the GPU": 6}, "gapped": {"done on the GPU": 6}}),
    ("axt", [T + "pseudocat.fa", T + "pseudopig.fa", "--format=axt"], "base_test.default.axt", "axt", None,
     {"table": {"built on the GPU": 1}, "search": {"done on the GPU": 6}, "gapped": {"done on the GPU": 6}}),
    ("chained", [T + "pseudocat.fa", T + "pseudopig.fa", "C=1", "W=8", "T=0"], "base_test.chained.lav", "lav", None,
     {"table": {"built on the GPU": 1}, "search": {"done on the GPU": 6}, "chain": {"done by the library": 6}}),
    ("extended", [T + "pseudocat.fa", T + "pseudopig.fa", "C=2", "W=8", "T=0"], "base_test.extended.lav", "lav", None,
     {"table": {"built on the GPU": 1}, "search": {"done on the GPU": 6}, "gapped": {"done on the GPU": 6}}),
    ("interpolated", [T + "pseudocat.fa", T + "pseudopig.fa", "C=2", "W=8", "T=0", "H=2200"], "base_test.interpolated.lav", "lav", None,
     # the tweener's 15 inner windows (7-mer tables on <= 20 kbp, src/tweener.c:769-829): searched and extended as
     # batches, one of each per (query, strand) -- lzgpu_window_search, lzgpu_gapped_extend_batch
     {"table": {"built on the GPU": 1}, "search": {"done on the GPU": 6}, "gapped": {"done on the GPU": 6},
      "tweener": {"windows searched on the GPU": 6, "windows extended on the GPU": 6, "done on the GPU": 6}}),
    ("stdin2", [T + "pseudocat.fa", "C=3", "W=8", "T=0"], None, None, "pseudopig.fa",
     {"table": {"built on the GPU": 1}, "search": {"done on the GPU": 6}}),
    ("2bit1", [T + "pseudopig.2bit/pig2", T + "pseudocat.fa", "C=2", "W=8", "T=0"], None, None, None,
     {"table": {"built on the GPU": 1}, "search": {"done on the GPU": 2}, "gapped": {"done on the GPU": 2}}),
    ("2bit2", [T + "pseudocat.fa", T + "pseudopig.2bit", "C=2", "W=8", "T=0"], None, None, None,
     {"table": {"built on the GPU": 1}, "search": {"done on the GPU": 6}, "gapped": {"done on the GPU": 6}}),
    ("seeded", [T + "pseudocat.fa", T + "pseudopig.fa", "C=3", "--seed=111010011101"], "base_test.seeded.lav", "lav", None,
     {"table": {"built on the GPU": 1}, "search": {"done on the GPU": 6}}),
    ("hw_seeded", [T + "pseudocat.fa", T + "pseudopig.fa", "C=3", "--seed=TTT0T0T0TTT00T0T"], "base_test.hwseeded.lav", "lav", None,
     {"table": {"reference path": 1}, "search": {"reference path": 6}}),                      # half-weight seed: declined
    ("ow_seeded", [T + "pseudocat.fa", T + "pseudopig.fa", "--justhits", "--seed=111010011101", "--word=12", "--gfa"], None, None, None,
     {"table": {"reference path": 1}, "search": {"reference path": 6}}),                      # overweight seed (--word): declined
    # dynamic masking rewrites target bases in place and takes their seeds out of the table (mask_seed_position_table):
    # the device's table is copied to the host and dropped at the first masked interval, its copy of the target
    # bytes as well (re-uploaded for the next gapped stage) -- the case that found a stale device target in round 3
    ("masking", [T + "fake_apple.fa", T + "fake_orange_reads.fa", "--masking=3"], "base_test.masking.lav", "lav", None,
     {"table": {"built on the GPU": 1, "copied to the host for a reference routine": 1}, "search": {"done on the GPU": 13, "reference path": 187},
      "gapped": {"done on the GPU": 78, "reference path": 122}}),
    ("anchors", [T + "aglobin.2bit/human", T + "aglobin.2bit/cow", "C=0", "--format=maf-", "--anchors=" + T + "base_test.anchors.anchors"],
     "base_test.anchors.maf", "diff", None, {"gapped": {"done on the GPU": 1, "reference path": 1}}),   # (the strand without anchors: nothing to extend)
    ("anchors_multi", [T + "aglobin.2bit/human", T + "shorties.fa[subset=" + T + "shorties.names]", "C=0", "--format=maf-",
                       "--anchors=" + T + "base_test.anchors_multi.anchors"], "base_test.anchors_multi.maf", "diff", None,
     {"gapped": {"done on the GPU": 9, "reference path": 11}}),
    ("subrange", [T + "aglobin.2bit/human[10000,60000]", T + "aglobin.2bit/cow[15000#40000]"], "base_test.subrange.lav", "lav", None,
     {"table": {"built on the GPU": 1}, "search": {"done on the GPU": 2}, "gapped": {"done on the GPU": 1, "reference path": 1}}),
    ("mask", [T + "pseudocat.fa", T + "pseudopig.fa[nmask=" + T + "pseudopig.n.mask]", "--ambiguous=n,60"], "base_test.mask.lav", "lav", None,
     {"table": {"built on the GPU": 1}, "search": {"done on the GPU": 6}, "gapped": {"done on the GPU": 6}}),
    ("coi_fa", [T + "aglobin.2bit/human", T + "shorties.fa[subset=" + T + "shorties.names]", "K=3000", "--maf-"], "base_test.coi.maf", "diff", None,
     {"table": {"built on the GPU": 1}, "search": {"done on the GPU": 20}, "gapped": {"done on the GPU": 9, "reference path": 11}}),
    ("coi_2bit", [T + "aglobin.2bit/human", T + "shorties.2bit[subset=" + T + "shorties.names]", "K=3000", "--maf-"], "base_test.coi.maf", "diff", None,
     {"table": {"built on the GPU": 1}, "search": {"done on the GPU": 20}, "gapped": {"done on the GPU": 9, "reference path": 11}}),
    ("multi", [T + "aglobin.2bit/human", T + "shorties.2bit[multi,@" + T + "shorties.names]", "K=3000", "--maf-"], None, None, None,
     {"table": {"built on the GPU": 1}, "search": {"done on the GPU": 2}, "gapped": {"done on the GPU": 2}}),        # [multi]: one partitioned query
    ("multi_subrange", [T + "aglobin.2bit/human", T + "shorties.2bit[multi,51..200]", "K=3000", "--maf-"], "base_test.multi_subrange.maf", "diff", None,
     {"table": {"built on the GPU": 1}, "search": {"done on the GPU": 2}, "gapped": {"done on the GPU": 2}}),
]


@needs_bins
@pytest.mark.parametrize("name,args,golden,how,stdin,expect", CASES, ids=[c[0] for c in CASES])
def test_base_test(sandbox, name, args, golden, how, stdin, expect):
    feed = open(sandbox / "test_data" / stdin).read() if stdin else None
    want, _ = run(REF_BIN, args, sandbox / "src", feed)
    got, err = run(GPU_BIN, args, sandbox / "src", feed, verbose=True)
    assert got == want                                           # the pristine binary, same box, same command line
    seen = notes(err)
    with open(os.path.join(os.environ.get("LZGPU_NOTES_DIR", str(sandbox)), "notes_%s.txt" % name), "w") as f:
        f.write(repr(seen) + "\n")
    if golden:
        g = open(os.path.join(H.GOLDEN, golden)).read()
        if how == "diff":
            assert got == g
        elif how == "lav":
            assert normalize_lav(got.replace("(stdin)", "../test_data/pseudopig.fa")) == normalize_lav(g)
        elif how == "axt":                                       # (tools/axt_compare.py ignores comment lines)
            strip = lambda s: [l for l in s.split("\n") if not l.startswith("#")]
            assert strip(got) == strip(g)
    if expect is not None:
        for stage, hows in expect.items():
            for how_, n in hows.items():
                assert seen.get(stage, {}).get(how_, 0) == n, (stage, seen)
    # whatever ran, nothing failed over silently: every stage line is one of the known outcomes
    known = {"built on the GPU", "done on the GPU", "reference path", "declined, reference path", "loaded from the table cache",
             "copied to the host for a reference routine", "unit of another rank", "shared with the other ranks", "received from rank 0",
             "windows searched on the GPU", "windows extended on the GPU", "no windows", "done by the library"}
    assert all(h in known for st in seen.values() for h in st), seen


@needs_bins
def test_segments_round_trip(sandbox):
    """base_test_segments (src/Makefile:384-400): HSPs written as segments, then the gapped stage alone from that file
    must give the default output -- B3 without B2"""
    args = [T + "pseudocat.fa", T + "pseudopig.fa"]
    hsps, _ = run(GPU_BIN, args + ["--nogapped", "--format=segments"], sandbox / "src", verbose=True)
    ref_hsps, _ = run(REF_BIN, args + ["--nogapped", "--format=segments"], sandbox / "src")
    assert hsps == ref_hsps
    with open(sandbox / "test_results" / "base_test.segments.hsps", "w") as f:
        f.write(hsps)
    out, err = run(GPU_BIN, args + ["--segments=../test_results/base_test.segments.hsps"], sandbox / "src", verbose=True)
    assert normalize_lav(out) == normalize_lav(open(os.path.join(H.GOLDEN, "base_test.default.lav")).read())
    assert notes(err).get("gapped", {}).get("done on the GPU", 0) == 6
