// lz_chain_host.cpp -- N2 of SURVEY.md 8(f): reduce_to_chain (src/chain.c:497), the reference's chaining of the HSPs
// of a (query, strand) before the gapped stage (--chain).  Host code: the recurrence
//     chain[i] = scale * s_i + max(0, max_{j before i in both sequences} (chain[j] - connect(j, i)))
// runs over the anchors in pos1 order and every step needs the finished values of all earlier anchors -- 79 k dependent
// steps for a 50 Mbp strand, each a search in a 2-d tree -- which no device formulation shortens.  What is
// MI355X-specific about it is where it sits: behind the same C ABI as B1-B3, so that the bound lastz and bench.py
// take it from the library, next to the stages it feeds.
//
// Bit-exactness is a matter of reproducing WHICH predecessor wins, ties included.  The reference searches a K-d tree
// over (diagonal, pos2) depth first and keeps the first candidate that is strictly better than what it has, so the
// winner among equal chain scores is the first in ITS traversal order -- a property of its tree.  The tree is therefore
// built as the reference builds it (median-of-three pivots, its partition loop, buckets of three: src/chain.c:626-666,
// :805-862), and walked in its order with its pruning test (:920-990) -- including the two calls of :960-961 that pass
// (lowerBound, 1 - axis) where the signature says (axis, lowerBound): below a node cut by pos2 the "axis" of the
// children is (int) lowerBound and their lower bound is 1 - axis.  That slip decides which subtrees a query enters, so
// it is part of the function computed here.  Doubles (the reference's `bigscore`), its comparison operators, its
// integer widths (sgnpos and score are 32-bit in the default build) are kept.
//
// The layout is this file's own: the tree is an array of nodes in construction order (children by index), the
// permutation and the chain scores are flat vectors, the connection penalty -- the reference passes a callback,
// chain_connect_penalty (src/lastz.c:3687-3741) -- is that function's closed form over three constants.
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include <thread>
#include <atomic>
#include <system_error>
#include "lz_common.hpp"
#include "../../include/lzgpu.h"

int lz_fail(int code, const char* fmt, ...);

namespace {
typedef double bigscore;                                      // src/chain.c:60

struct Node { u32 lo, hi; s32 cut; bigscore best; s32 lo_son, hi_son; bool bucket; };   // lo_son / hi_son: indices, -1 = none
const u32 kBucket = 3;                                        // src/chain.c:110

struct Chain {
    const lz_segment* seg; u32 n;                             // the anchors in pos1 order (qSegmentsByPos1)
    const lz_chain_args* a;
    std::vector<u32> perm, inv;
    std::vector<bigscore> score;
    std::vector<Node> nodes;
    // the query of the search in flight
    u32 qi, x, y; s32 diag;

    s32 proj(u32 i, int axis) const                            // projection(), src/chain.c:117-119 (sgnpos = s32: wraps like the reference)
    { const lz_segment& s = seg[perm[i]]; return axis == 0 ? (s32)((u32)s.pos1 - (u32)s.pos2) : (s32)s.pos2; }
    void swap(u32 p, u32 q) { const u32 t = perm[p]; perm[p] = perm[q]; perm[q] = t; }

    // partition_segments, src/chain.c:805-862: the pivot's final place m with lo..m-1 <= m < m+1..hi
    u32 partition(u32 lo, u32 hi, int axis)
    {
        for (;;) {
            const u32 m = (lo + hi) / 2;
            const s32 pa = proj(lo, axis), pb = proj(m, axis), pc = proj(hi, axis);
            s32 pivot;
            if ((pa <= pb && pb <= pc) || (pc <= pb && pb <= pa)) { swap(lo, m); pivot = pb; }
            else if ((pa <= pc && pc <= pb) || (pb <= pc && pc <= pa)) { swap(lo, hi); pivot = pc; }
            else pivot = pa;
            u32 i = lo, j = hi + 1;
            while (i < j) {
                for (i++; i <= hi && proj(i, axis) <= pivot; i++) ;
                for (j--; j >= lo && proj(j, axis) > pivot; j--) ;      // (proj(lo) == pivot stops it at lo)
                swap(i, j);
            }
            swap(i, j);                                          // undo the last swap
            swap(lo, j);                                         // the pivot to its place
            if (j < hi) return j;
            if (hi - lo == 2) return hi - 1;
            hi--;                                                // the pivot was the maximum: again without it
        }
    }
    s32 build(u32 lo, u32 hi, int axis)                          // build_kd_tree, src/chain.c:626-666
    {
        const s32 id = (s32)nodes.size();
        nodes.push_back(Node());
        nodes[id].best = 0; nodes[id].lo_son = nodes[id].hi_son = -1;
        if (hi + 1 - lo <= kBucket) { nodes[id].bucket = true; nodes[id].lo = lo; nodes[id].hi = hi; nodes[id].cut = 0; return id; }
        const u32 m = partition(lo, hi, axis);
        nodes[id].bucket = false; nodes[id].cut = proj(m, axis); nodes[id].hi = m; nodes[id].lo = 0;
        const s32 l = build(lo, m, 1 - axis);
        const s32 h = build(m + 1, hi, 1 - axis);
        nodes[id].lo_son = l; nodes[id].hi_son = h;
        return id;
    }

    // chain_connect_penalty, src/lastz.c:3687-3741, for a predecessor s of the query q (s starts before q in both)
    s32 connect(const lz_segment& s, const lz_segment& q) const
    {
        const u32 x_end = s.pos1 + s.length - 1, y_end = s.pos2 + s.length - 1;
        const s32 d1 = (s32)((u32)s.pos1 - (u32)s.pos2), d2 = (s32)((u32)q.pos1 - (u32)q.pos2);
        s32 dd = (s32)((u32)d2 - (u32)d1), subs;
        if (dd >= 0) subs = (s32)((u32)q.pos2 - y_end - 1u);
        else { subs = (s32)((u32)q.pos1 - x_end - 1u); dd = (s32)(0u - (u32)dd); }
        bigscore pen = (bigscore)(s32)((u32)dd * (u32)a->chain_diag);                    // (int products, as the reference computes them)
        if (subs >= 0) pen += (bigscore)(s32)((u32)subs * (u32)a->chain_anti);
        else           pen += (bigscore)(s32)((0u - (u32)subs) * (u32)a->scale * (u32)a->overlap_sub);
        if (pen > (bigscore)0x7FFFFFFF) return 0x7FFFFFFF;
        return (s32)pen;
    }

    struct Best { u32 num; bigscore contrib; };
    // best_predecessor, src/chain.c:920-990.  `axis` and `lower` are the values the reference's calls pass in these
    // two positions -- see the head of this file for :960-961.
    void search(s32 id, int axis, bigscore lower, Best& bp) const
    {
        const Node& nd = nodes[id];
        if (bp.contrib >= nd.best - lower) return;
        if (nd.bucket) {
            for (u32 i = nd.lo; i <= nd.hi; i++) {
                const u32 j = perm[i];
                const lz_segment& s = seg[j];
                if (s.pos1 >= x || s.pos2 >= y) continue;
                const bigscore p = score[j] - (bigscore)connect(s, seg[qi]);
                if (p > bp.contrib) { bp.contrib = p; bp.num = j; }
            }
        } else if (axis == 1) {
            // (sic: the children's "axis".  The reference's default build converts with x86's cvttsd2si, which yields INT_MIN for
            // every double outside int's range -- a conversion that is undefined in C++; spelled out here so that another compiler
            // or target picks the same child: ADVICE r4)
            const int ax = (lower >= -2147483648.0 && lower < 2147483648.0) ? (int)lower : (int)0x80000000;
            const bigscore lb = (bigscore)(int)(1u - (unsigned)axis);   // (sic: their lower bound)
            if ((s32)y >= nd.cut) search(nd.hi_son, ax, lb, bp);
            search(nd.lo_son, ax, lb, bp);
        } else {
            const int ax = (int)(1u - (unsigned)axis);
            const bigscore diff = (bigscore)(s32)((u32)diag - (u32)nd.cut);
            if (diff >= 0) { search(nd.hi_son, ax, lower, bp); search(nd.lo_son, ax, diff * (bigscore)a->diag_pen, bp); }
            else           { search(nd.lo_son, ax, lower, bp); search(nd.hi_son, ax, -diff * (bigscore)a->anti_pen, bp); }
        }
    }
    void propagate(bigscore s, u32 ix)                           // propagate_max_score, src/chain.c:1010-1023
    {
        for (s32 id = 0; id >= 0; ) {
            Node& nd = nodes[id];
            if (s > nd.best) nd.best = s;
            id = (ix <= nd.hi) ? nd.lo_son : nd.hi_son;
        }
    }
};

// qSegmentsByPos1, src/segment.c:1657-1680 (a total order up to identical records)
bool by_pos1(const lz_segment& p, const lz_segment& q)
{
    if (p.pos1 != q.pos1) return p.pos1 < q.pos1;
    if (p.length != q.length) return p.length < q.length;
    if (p.pos2 != q.pos2) return p.pos2 < q.pos2;
    if (p.id != q.id) return p.id < q.id;
    return p.s < q.s;
}
}

extern "C" int lzgpu_reduce_to_chain(const lz_chain_args* a, const lz_segment* segs, uint32_t n, uint32_t** kept, uint32_t* n_kept, int32_t* best_out)
{
    if (!a || (!segs && n) || !kept || !n_kept) return lz_fail(LZGPU_ERR_ARG, "lzgpu_reduce_to_chain: null argument");
    *kept = nullptr; *n_kept = 0;
    if (best_out) *best_out = 0;
    if (n == 0) return 0;
    if (a->scale == 0) return LZGPU_NH_UNSUPPORTED;
    // the anchors in the reference's order, remembering where each came from
    std::vector<u32> order(n);
    for (u32 i = 0; i < n; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](u32 p, u32 q) { return by_pos1(segs[p], segs[q]) || (!by_pos1(segs[q], segs[p]) && p < q); });
    std::vector<lz_segment> sorted(n);
    for (u32 i = 0; i < n; i++) sorted[i] = segs[order[i]];

    Chain c; c.seg = sorted.data(); c.n = n; c.a = a;
    c.perm.assign((size_t)n + 1, 0);                              // (one past the end: the partition loop's last swap touches it and puts it back)
    c.inv.resize(n); c.score.assign(n, 0.0);
    for (u32 i = 0; i < n; i++) c.perm[i] = i;
    c.nodes.reserve((size_t)n);
    c.build(0, n - 1, 1);
    for (u32 i = 0; i < n; i++) c.inv[c.perm[i]] = i;

    std::vector<u32> pred(n);
    bigscore best = 0; u32 best_end = 0xFFFFFFFFu;
    for (u32 i = 0; i < n; i++) {
        c.qi = i; c.x = sorted[i].pos1; c.y = sorted[i].pos2; c.diag = (s32)((u32)c.x - (u32)c.y);
        Chain::Best bp = { 0xFFFFFFFFu, 0.0 };
        c.search(0, 1, 0.0, bp);
        c.score[i] = (bigscore)sorted[i].s * (bigscore)a->scale + bp.contrib;
        if (c.score[i] > best) { best = c.score[i]; best_end = i; }
        pred[i] = bp.num;
        c.propagate(c.score[i], c.inv[i]);
    }
    // the chain, back from its end; what the reference leaves in the table: its members, still in pos1 order
    std::vector<u8> in(n, 0);
    u32 cnt = 0;
    for (u32 i = best_end; i != 0xFFFFFFFFu; i = pred[i]) { in[i] = 1; cnt++; }
    u32* out = (u32*)malloc((cnt ? cnt : 1) * sizeof(u32));
    if (!out) return lz_fail(LZGPU_ERR_OOM, "host malloc failed");
    u32 w = 0;
    for (u32 i = 0; i < n; i++) if (in[i]) out[w++] = order[i];
    *kept = out; *n_kept = cnt;
    if (best_out) {                                              // src/chain.c:598-606 (integer scores: rounded, clipped)
        bigscore b = (best / (bigscore)a->scale) + 0.5;
        if (b > (bigscore)0x7FFFFFFF) b = (bigscore)0x7FFFFFFF;
        *best_out = (s32)b;
    }
    return 0;
}


// Several independent chaining problems -- the two strands of a query, the units a rank has finished searching -- at once, one host
// thread each (at most `LZGPU_CHAIN_THREADS`, default 16).  A problem is serial by nature (every anchor needs the finished scores of all
// earlier ones); what need not be serial is one problem after the other: --chain on a 200 Mbp pair is 90 ms per strand, and the two
// strands' chains -- or a unit's chain and the next unit's search on the device -- do not wait for each other.
// Results are those of k calls of lzgpu_reduce_to_chain; the return code is that of the first problem that did not return 0.
extern "C" int lzgpu_reduce_to_chain_batch(const lz_chain_args* a, const lz_segment* const* segs, const uint32_t* n, uint32_t k,
                                           uint32_t** kept, uint32_t* n_kept, int32_t* best)
{
    if (!a || !segs || !n || !kept || !n_kept || k == 0) return lz_fail(LZGPU_ERR_ARG, "lzgpu_reduce_to_chain_batch: null argument");
    for (u32 j = 0; j < k; j++) { kept[j] = nullptr; n_kept[j] = 0; if (best) best[j] = 0; }
    static const u32 cap = []() { const char* e = getenv("LZGPU_CHAIN_THREADS"); const int v = e ? atoi(e) : 0; return (u32)(v > 0 ? v : 16); }();
    std::vector<int> rcs(k, 0);
    std::atomic<u32> next{0};
    auto work = [&]() {
        for (;;) {
            const u32 j = next.fetch_add(1);
            if (j >= k) break;
            rcs[j] = lzgpu_reduce_to_chain(a, segs[j], n[j], &kept[j], &n_kept[j], best ? &best[j] : nullptr);
        }
    };
    {
        std::vector<std::thread> th;
        const u32 want = k < cap ? k : cap;
        try { for (u32 t = 1; t < want; t++) th.emplace_back(work); }
        catch (const std::system_error&) { /* fewer helpers than asked for: the ones that started, and this thread, do the work */ }
        work();
        for (auto& t : th) t.join();
    }
    for (u32 j = 0; j < k; j++)
        if (rcs[j]) {
            for (u32 i = 0; i < k; i++) { free(kept[i]); kept[i] = nullptr; n_kept[i] = 0; }
            return rcs[j];
        }
    return 0;
}
