"""BASELINE.json configs[3] and [4] in their shape: one lastz process per GPU rank (here: two ranks on the one GPU
of the test box, table handed over through lzgpu_table_share's file transport because RCCL refuses two ranks on
one device), (query sequence x strand) units dealt out by LPT, outputs merged in the reference's order -- byte for
byte the single-process output and the pristine reference's; with --chain --inner=2000 --scores=HOXD70 on top
(configs[4]).  The RCCL transport itself is exercised with one rank (unique id, communicator, ncclBroadcast on the
library's stream)."""
import os
import subprocess
import pytest

from lastz_amd import seqio, multi
import helpers as H
from lavparse import normalize_lav

pytestmark = pytest.mark.gpu
GPU_BIN = os.path.join(H.ROOT, "integration", "_build", "lastz_gpu")
REF_BIN = os.path.join(H.ROOT, "oracle", "_ref", "lastz")
HOXD70 = os.path.join(H.ROOT, "lastz_amd", "data", "HOXD70.q")
needs_bins = pytest.mark.skipif(not (os.path.exists(GPU_BIN) and os.path.exists(REF_BIN)), reason="oracle/_ref binaries not built")


@pytest.fixture(scope="module")
def pair(tmp_path_factory):
    d = tmp_path_factory.mktemp("multi")
    t, q = seqio.synth_pair(1_500_000, 1_400_000, seed=73)
    _, q2 = seqio.synth_pair(1_500_000, 400_000, seed=74)
    seqio.write_fasta(d / "t.fa", [("target", t)])
    seqio.write_fasta(d / "q.fa", [("qa", q[:600_000]), ("qb", q2), ("qc", q[600_000:])])     # three query sequences
    return d


def _run(binary, args, cwd, env_extra=None):
    env = dict(os.environ); env.update(env_extra or {})
    p = subprocess.run([binary] + args, cwd=cwd, capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    return p.stdout, p.stderr


CONFIGS = {"configs3_gapped": ["--ydrop=9430"],
           "configs4_chain_inner_scores": ["--chain", "--inner=2000", "--scores=" + HOXD70]}


@needs_bins
@pytest.mark.parametrize("name", list(CONFIGS))
def test_two_ranks_merge_to_the_single_process_and_reference_output(pair, name):
    flags = CONFIGS[name]
    t, q = str(pair / "t.fa"), str(pair / "q.fa")
    merged, errs, plan = multi.run(t, q, flags, ranks=2, lastz=GPU_BIN, devices=[0, 0], transport="file",
                                   env={"LZGPU_VERBOSE": "1"})
    single, err1 = _run(GPU_BIN, [t, q] + flags, pair, {"LZGPU_VERBOSE": "1"})
    ref, _ = _run(REF_BIN, [t, q] + flags, pair)
    assert normalize_lav(merged) == normalize_lav(single)
    assert normalize_lav(merged) == normalize_lav(ref)
    assert merged.count("\na {") > 20
    # both ranks worked, each on its own units, and the hot path ran on the GPU
    assert sorted(u for p in plan for u in p) == [(i, s) for i in range(3) for s in (0, 1)] and all(plan)
    assert "[lzgpu] table: shared with the other ranks" in errs[0] and "[lzgpu] table: received from rank 0" in errs[1]
    assert multi.run.last["devices_bound"] == [0, 0]             # the start-up self-check: every rank printed the device the launcher gave it
    for r in (0, 1):
        assert "[lzgpu] rank %d of 2: device 0" % r in errs[r]
        assert errs[r].count("[lzgpu] search: done on the GPU") == len(plan[r])
        loaded = len({qi for qi, _ in plan[r]})                  # a rank reads only the sequences it owns a strand of
        assert errs[r].count("[lzgpu] search: unit of another rank") == 2 * loaded - len(plan[r])
        assert errs[r].count("contains an empty sequence") == 3 - loaded
        assert "[lzgpu] gapped: done on the GPU" in errs[r]
    assert "[lzgpu] gapped: done on the GPU" in err1


@needs_bins
def test_rccl_transport_with_one_rank(pair):
    """librccl loaded on demand, unique id through the rendezvous directory, communicator, one ncclBroadcast per
    table buffer on the library's stream -- all of lzgpu_table_share's RCCL path that a one-GPU box can run"""
    share = pair / "share1"; os.makedirs(share, exist_ok=True)
    out, err = _run(GPU_BIN, [str(pair / "t.fa"), str(pair / "q.fa"), "--nogapped"], pair,
                    {"LZGPU_VERBOSE": "1", "LZGPU_SHARE_FORCE": "1", "LZGPU_SHARE_DIR": str(share), "LZGPU_RANK": "0", "LZGPU_WORLD": "1"})
    ref, _ = _run(REF_BIN, [str(pair / "t.fa"), str(pair / "q.fa"), "--nogapped"], pair)
    assert "[lzgpu] table: shared with the other ranks" in err
    assert os.path.exists(share / "table0" / "nccl_id") and os.path.exists(share / "table0" / "geom")
    assert normalize_lav(out) == normalize_lav(ref)


@needs_bins
def test_a_rank_parses_only_what_it_owns_at_size(tmp_path):
    """configs[3] at a size where host time shows: a 200 Mbp target, two query sequences of 40 Mbp, two ranks (one
    GPU, file transport).  Each rank gets the query file with the other rank's record reduced to its header: it parses
    and reverse-complements 40 Mbp instead of 80, the table is never copied to the host (no reference routine reads
    it), and the merged LAV is the single-process LAV byte for byte."""
    import time
    t, qa = seqio.synth_pair(200_000_000, 40_000_000, seed=301)
    _, qb = seqio.synth_pair(200_000_000, 40_000_000, seed=302)
    seqio.write_fasta(tmp_path / "t.fa", [("target", t)])
    seqio.write_fasta(tmp_path / "q.fa", [("qa", qa), ("qb", qb)])
    tf, qf = str(tmp_path / "t.fa"), str(tmp_path / "q.fa")
    env = {"LZGPU_VERBOSE": "1", "LZGPU_VERBOSE_CLOCK": "1"}
    merged, errs, plan = multi.run(tf, qf, ["--nogapped"], ranks=2, lastz=GPU_BIN, devices=[0, 0], transport="file", env=env)
    info = dict(multi.run.last)
    t0 = time.time()
    single, err1 = _run(GPU_BIN, [tf, qf, "--nogapped"], tmp_path, {"LZGPU_VERBOSE": "1"})
    single_s = time.time() - t0
    assert merged == single                                     # byte for byte, d-stanza included
    assert info["split"] and info["owned_bases"] == [40_000_000, 40_000_000]
    assert sorted(plan) == [[(0, 0), (0, 1)], [(1, 0), (1, 1)]]
    for r in (0, 1):
        assert errs[r].count("[lzgpu] search: done on the GPU") == 2 and "unit of another rank" not in errs[r]
        assert "copied to the host for a reference routine" not in errs[r]
        assert errs[r].count("contains an empty sequence") == 1
    print("single process %.1f s; ranks %s s for %s owned bases" % (single_s, ["%.1f" % x for x in info["rank_seconds"]], info["owned_bases"]))
    # (two ranks share ONE GPU here and the table travels through a file instead of xGMI, so the wall clocks say
    # nothing about scaling: a loose bound against pathologies only)
    assert max(info["rank_seconds"]) < 3.0 * single_s


@needs_bins
@pytest.mark.parametrize("fmt", ["maf", "axt", "general:name1,zstart1,end1,name2,strand2,zstart2,end2,score,cigarx", "cigar", "differences", "sam", "softsam"])
def test_two_ranks_merge_line_oriented_formats(pair, fmt):
    """the records of MAF / AXT / SAM / general / cigar / differences output, put back in file order by the unit markers
    (integration/lzgpu_shim.c::unit_marker), are the single process's and the pristine reference's byte for byte"""
    flags = ["--ydrop=9430", "--format=" + fmt]
    t, q = str(pair / "t.fa"), str(pair / "q.fa")
    merged, errs, plan = multi.run(t, q, flags, ranks=2, lastz=GPU_BIN, devices=[0, 0], transport="file",
                                   env={"LZGPU_VERBOSE": "1"})
    ref, _ = _run(REF_BIN, [t, q] + flags, pair)
    drop = lambda s: "".join(l for l in s.splitlines(True) if not (l.startswith("#")) or l.startswith("#name"))   # header comments carry the binary's name
    assert "#lzgpu-unit" not in merged
    assert drop(merged) == drop(ref)
    assert len(drop(merged).splitlines()) > 20
    for r in (0, 1):
        assert "[lzgpu] gapped: done on the GPU" in errs[r]


def _bench(args, nproc, port, env_extra=None, timeout=900):
    import json, sys
    env = dict(os.environ); env.update({"LZ_BENCH_BACKEND": "gloo", "MASTER_ADDR": "127.0.0.1"}); env.update(env_extra or {})
    cmd = [sys.executable, os.path.join(H.ROOT, "bench.py")] if nproc == 1 else \
          [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(H.ROOT, "bench.py")]
    p = subprocess.run(cmd + args, capture_output=True, text=True, env=env, timeout=timeout, cwd=H.ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.split("\n") if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


SHAPE = ["--steps", "1", "--warmup", "0", "--tlen-multi", "4000000", "--q-units", "3", "--q-unit-len", "1500000", "--no-cpu-baseline"]


@needs_bins
def test_a_rank_that_is_not_on_the_device_it_was_given_stops_the_run(pair):
    """the self-check's other half: a launcher that hands rank 1 a device the box does not have (LOCAL_RANK=1 wraps to device 0 here)
    must not get a quiet run on the wrong GPU -- the shim stops the rank, the launcher the job"""
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs a box with exactly one visible device")
    t, q = str(pair / "t.fa"), str(pair / "q.fa")
    with pytest.raises(RuntimeError) as e:
        multi.run(t, q, ["--nogapped"], ranks=2, lastz=GPU_BIN, devices=[0, 1], transport="file", env={"LZGPU_SHARE_TIMEOUT_S": "20"})
    assert "LOCAL_RANK=1" in str(e.value) or "rank 1" in str(e.value)


def test_bench_two_ranks_search_and_gapped_stage_equal_one_rank():
    """bench.py's N > 1 code path: two ranks (gloo stands in for RCCL, which refuses two ranks on one
    device), every unit searched AND gapped-extended, B3 of unit k beside B2 of unit k+1 on a second host thread and
    stream -- the merged HSPs and alignments are those of one rank doing all units."""
    two = _bench(["--gpus", "2"] + SHAPE, 2, 29541)
    one = _bench(["--gpus", "1", "--force-multi"] + SHAPE, 1, 0)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["hsps_merged"] == one["hsps_merged"] > 1000
    assert two["alignments"] == one["alignments"] > 20
    assert two["alignments_sha"] == one["alignments_sha"]
    assert two["units_per_rank"] == [3, 3] and not two["bucket_owners"]
    kinds = {k for _, k, _, _ in two["timeline_rank0_last_step"]}
    assert kinds == {"search", "gapped"}
    for r in two["per_rank"]:
        assert r["search_s"] > 0 and r["gapped_s"] > 0
        assert 1 <= r["b3_batches"] <= 2                        # three units of a rank: the strands of a sequence go down together (VERDICT r4 #5b)
    assert 0 < two["table_build_and_broadcast_share_of_step"] < 1
    # start-up self-check (VERDICT r4 #5c): every rank's library on the device of its LOCAL_RANK; here the two ranks share the one device by design
    for line in (one, two):
        sc = line["device_selfcheck"]
        assert sc["ok"] and all(e["lzgpu_device_index"] == e["local_rank"] for e in sc["ranks"])
    assert two["device_selfcheck"]["ranks_share_a_device"] or two["device_selfcheck"]["devices_visible"] >= 2


def test_bench_multi_path_over_rccl_with_one_rank():
    """bench.py's N > 1 branch on its DEFAULT backend -- torch.distributed "nccl" = RCCL, world = 1 -- so that
    init_process_group("nccl", device_id=...) and bcast_table's zero-copy RCCL broadcast of the library's own device buffers have run
    before the driver's 8-GPU node runs them; rank 0's stdout is exactly ONE line (the JSON); same results as the gloo yardstick."""
    import json, sys
    env = {k: v for k, v in os.environ.items() if k != "LZ_BENCH_BACKEND"}
    env["MASTER_ADDR"] = "127.0.0.1"
    p = subprocess.run([sys.executable, os.path.join(H.ROOT, "bench.py"), "--gpus", "1", "--force-multi"] + SHAPE,
                       capture_output=True, text=True, env=env, timeout=900, cwd=H.ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    out_lines = p.stdout.split("\n")
    assert len(out_lines) == 2 and out_lines[1] == "" and out_lines[0].startswith("{"), p.stdout[:2000]      # one line, nothing ahead of it
    rccl = json.loads(out_lines[0])
    assert rccl["table_transport"] == "rccl" and rccl["table_bytes"] > 64 << 20
    assert rccl["n_gpus"] == 1 and rccl["lpt_imbalance"] == 1.0 and [r["units"] for r in rccl["per_rank"]] == [6]
    gloo = _bench(["--gpus", "1", "--force-multi"] + SHAPE, 1, 0)
    assert gloo["table_transport"].startswith("gloo")
    assert rccl["hsps_merged"] == gloo["hsps_merged"] > 1000
    assert rccl["alignments_sha"] == gloo["alignments_sha"] and rccl["alignments"] == gloo["alignments"] > 20


def test_bench_chain_in_front_of_the_gapped_stage_on_the_b3_thread():
    """configs[4]'s shape: --chain -- every unit's HSPs chained (lzgpu_reduce_to_chain_batch, host code on the B3 thread beside the next
    unit's search), the chains extended; two ranks = one rank, and the chained job extends fewer anchors into fewer alignments"""
    two = _bench(["--gpus", "2", "--chain"] + SHAPE, 2, 29545)
    one = _bench(["--gpus", "1", "--force-multi", "--chain"] + SHAPE, 1, 0)
    plain = _bench(["--gpus", "1", "--force-multi"] + SHAPE, 1, 0)
    assert two["chain"] and one["chain"] and not plain["chain"]
    assert two["alignments_sha"] == one["alignments_sha"] != plain["alignments_sha"]
    assert 0 < one["alignments"] <= plain["alignments"]
    assert one["hsps_merged"] == plain["hsps_merged"]
    assert all(r["chain_s_inside_gapped_s"] > 0 for r in two["per_rank"])


def test_bench_bucket_owners_inside_the_units():
    """fewer units than ranks: B2 sharded inside every unit by hashed-diagonal ownership, the parts merged on the rank
    that runs the unit's gapped stage -- same HSPs, same alignments as whole units on one rank"""
    shape = ["--steps", "1", "--warmup", "0", "--tlen-multi", "4000000", "--q-units", "1", "--q-unit-len", "2000000", "--no-cpu-baseline"]
    own = _bench(["--gpus", "2", "--bucket-owners", "always"] + shape, 2, 29543)
    one = _bench(["--gpus", "1", "--force-multi"] + shape, 1, 0)
    assert own["bucket_owners"] and not one["bucket_owners"]
    assert own["hsps_merged"] == one["hsps_merged"] > 500
    assert own["alignments_sha"] == one["alignments_sha"]


def test_a_rank_on_the_second_device_binds_every_thread_to_it():
    """HIP's current device is per host thread.  With LOCAL_RANK=1 every allocation and launch of the
    library -- the caller's thread, lzgpu_init_async's, the gapped batch's workers -- must be on device 1."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one visible device")
    code = ("import numpy as np, torch\n"
            "from lastz_amd import lzgpu, seqio\n"
            "from oracle import lzo\n"
            "lib = lzgpu.Lib(); lib.L.lzgpu_init_async(-1); lib.init(-1)\n"
            "assert lib.L.lzgpu_device_index() == 1\n"
            "t, q = seqio.synth_pair(400000, 400000, seed=5)\n"
            "free0 = torch.cuda.mem_get_info(0)[0]\n"
            "lib.table_prepare(t, lib.seed(), lzo.upper_nuc_to_bits())\n"
            "_, masked = lzo.hoxd70_scoring()\n"
            "hs = lib.seed_hit_search(masked, q=q)\n"
            "assert len(hs) > 10\n"
            "assert abs(torch.cuda.mem_get_info(0)[0] - free0) < (64 << 20), 'the library allocated on device 0'\n"
            "print('ok')\n")
    import sys
    env = dict(os.environ); env["LOCAL_RANK"] = "1"
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=H.ROOT, timeout=600)
    assert p.returncode == 0 and "ok" in p.stdout, p.stderr[-2000:]
