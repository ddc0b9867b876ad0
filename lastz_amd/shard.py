"""Multi-GPU sharding of the hot path.

The path shards exactly at (query sequence x strand) granularity: every seed_hit_search call starts
from an empty diagonal hash (src/seed_search.c:362) and every gapped_extend call from an empty
alignment list (src/gapped_extend.c:1051), so units are independent.  One process per GPU; the
target position table is built once (rank 0) and broadcast; there is no data-path collective.
Results are merged on rank 0 in the reference's output order: queries in file order, + strand
before - strand (src/lastz.c:1592-1691).
"""
from typing import List, Sequence, Tuple


def plan_units(lengths: Sequence[int], world: int, strands: int = 2) -> List[List[Tuple[int, int]]]:
    """Longest-processing-time assignment of (query index, strand) units to ranks; seed work is
    proportional to Tlen*Qlen, so the unit weight is the query length.  Deterministic."""
    units = [(i, s) for i in range(len(lengths)) for s in range(strands)]
    order = sorted(units, key=lambda u: (-lengths[u[0]], u[0], u[1]))
    load = [0] * world
    plan: List[List[Tuple[int, int]]] = [[] for _ in range(world)]
    for u in order:
        r = min(range(world), key=lambda k: (load[k], k))
        plan[r].append(u)
        load[r] += lengths[u[0]]
    for p in plan:
        p.sort()
    return plan


def merge_units(per_rank: Sequence[dict]) -> List[Tuple[Tuple[int, int], object]]:
    """per_rank[r] = {(query index, strand): result}; -> [(unit, result)] in file order, + before -."""
    out = {}
    for d in per_rank:
        for k, v in d.items():
            if k in out:
                raise ValueError(f"unit {k} computed twice")
            out[k] = v
    return sorted(out.items(), key=lambda kv: kv[0])


def merge_bucket_owners(per_owner):
    """Sharding inside one query by hashed-diagonal ownership (lzgpu_set_bucket_owner): per_owner[r] =
    (hsps, order) with order = the (n, 2) uint64 sort words of lzgpu_last_hsp_order.  Returns the HSPs in
    the single-process discovery order (query position, probe index, target position descending)."""
    import numpy as np
    hs = np.concatenate([h for h, _ in per_owner]) if per_owner else np.zeros(0)
    od = np.concatenate([o.reshape(-1, 2) for _, o in per_owner]) if per_owner else np.zeros((0, 2), dtype=np.uint64)
    if len(hs) == 0:
        return hs
    idx = np.lexsort((od[:, 1], od[:, 0]))                  # primary: word 0, then word 1; owners never tie
    return hs[idx]


def broadcast_buffers(dist, tensors, src: int = 0):
    """One broadcast per table buffer (target bytes, wstart, wpos).  On the GPU these are zero-copy
    views of the library's device allocations and the backend is RCCL over xGMI; the CPU tests
    run the same call over gloo."""
    for t in tensors:
        if t.numel():
            dist.broadcast(t, src=src)
