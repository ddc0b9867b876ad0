"""N>1 path on CPU: world_size-2 gloo processes.  Each rank runs its share of (query, strand) units
through the seed stage (the CPU emulation of the device pipeline stands in for the GPU here -- test
infrastructure), the table buffers travel by broadcast, rank 0 merges in the reference's order, and
the merged result must equal the single-process result."""
import os
import socket
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lastz_amd import shard


def test_plan_is_balanced_and_complete():
    lens = [200, 50, 50, 50, 120, 10, 10]
    plan = shard.plan_units(lens, 3)
    flat = sorted(u for p in plan for u in p)
    assert flat == sorted((i, s) for i in range(len(lens)) for s in (0, 1))
    loads = [sum(lens[i] for i, _ in p) for p in plan]
    assert max(loads) - min(loads) <= max(lens)
    assert shard.plan_units(lens, 3) == plan                      # deterministic
    assert shard.plan_units([5], 4) == [[(0, 0)], [(0, 1)], [], []]


def test_merge_order_and_duplicates():
    merged = shard.merge_units([{(1, 0): "c", (0, 1): "b"}, {(0, 0): "a", (1, 1): "d"}])
    assert [v for _, v in merged] == ["a", "b", "c", "d"]
    with pytest.raises(ValueError):
        shard.merge_units([{(0, 0): 1}, {(0, 0): 2}])


def _worker(rank, world, port, root, out_path):
    import sys
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import lzo
    from lastz_amd import seqio, lzgpu
    import helpers as H
    em = lzgpu.Lib(path=os.path.join(root, "tests", "emul", "libemul.so"), prefix="emul_")
    _, masked = H.scoring()
    ctb = lzo.upper_nuc_to_bits()
    t, q0 = H.load_case("synth_overlap")
    queries = [q0[:60000], q0[60000:100000], q0[100000:]]

    # rank 0 owns the target; the table buffers are broadcast (here: the oracle's CSR as stand-in buffers)
    if rank == 0:
        ws, wp = lzo.Table(t, lzo.seed()).csr()
        geom = [len(t), len(ws), len(wp)]
    else:
        geom = [0, 0, 0]
    g = torch.tensor(geom, dtype=torch.int64); dist.broadcast(g, src=0)
    tlen, nws, nwp = (int(x) for x in g)
    bufs = [torch.from_numpy(t.copy()) if rank == 0 else torch.empty(tlen, dtype=torch.uint8),
            torch.from_numpy(ws.view(np.int32)) if rank == 0 else torch.empty(nws, dtype=torch.int32),
            torch.from_numpy(wp.view(np.int32)) if rank == 0 else torch.empty(nwp, dtype=torch.int32)]
    shard.broadcast_buffers(dist, bufs, src=0)
    t_local = bufs[0].numpy()
    lws, lwp = lzo.Table(t_local, lzo.seed()).csr()
    assert (bufs[1].numpy().view(np.uint32) == lws).all() and (bufs[2].numpy().view(np.uint32) == lwp).all()

    em.table_prepare(t_local, em.seed(), ctb)
    plan = shard.plan_units([len(q) for q in queries], world)
    mine = {}
    for qi, strand in plan[rank]:
        qq = queries[qi] if strand == 0 else seqio.revcomp(queries[qi])
        mine[(qi, strand)] = em.seed_hit_search(masked, q=qq).tolist()
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(mine, gathered, dst=0)
    if rank == 0:
        merged = shard.merge_units(gathered)
        np.save(out_path, np.array([repr(merged)], dtype=object), allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()


def test_merge_bucket_owners_restores_discovery_order():
    rng = np.random.default_rng(5)
    n = 5000
    order = np.stack([rng.integers(0, 1 << 40, n, dtype=np.uint64) | np.uint64(0), rng.integers(0, 1 << 32, n, dtype=np.uint64)], axis=1)
    order = order[np.lexsort((order[:, 1], order[:, 0]))]
    hsps = np.arange(n)                                      # stand-in payload: the rank in discovery order
    owner = rng.integers(0, 3, n)
    per_owner = [(hsps[owner == r], order[owner == r]) for r in range(3)]
    assert (shard.merge_bucket_owners(per_owner) == hsps).all()
    assert len(shard.merge_bucket_owners([(hsps[:0], order[:0])])) == 0


def test_two_ranks_equal_one(tmp_path):
    from oracle import lzo
    from lastz_amd import seqio
    import helpers as H
    if not os.path.exists(os.path.join(H.ROOT, "tests", "emul", "libemul.so")):
        pytest.skip("emulation harness not built (tests/test_emul_vs_oracle.py builds it)")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "merged.npy")
    mp.spawn(_worker, args=(2, port, H.ROOT, out), nprocs=2, join=True)
    merged = np.load(out, allow_pickle=True)[0]
    _, masked = H.scoring()
    t, q0 = H.load_case("synth_overlap")
    queries = [q0[:60000], q0[60000:100000], q0[100000:]]
    tab = lzo.Table(t, lzo.seed())
    want = []
    for qi, q in enumerate(queries):
        for strand, qq in ((0, q), (1, seqio.revcomp(q))):
            h, _ = lzo.seed_hit_search(tab, qq, masked)
            want.append(((qi, strand), h.tolist()))
    assert merged == repr(want)
