// lz_gapped_host.cpp -- see lz_gapped_host.hpp.  Pure host code (no HIP calls); the one-sided
// DPs are delegated to an LzDpExecutor (the HIP executor in dp_kernels.hip).
#include <string.h>
#include <algorithm>
#include <thread>
#include <functional>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <unordered_map>
#include <deque>
#include "lz_gapped_host.hpp"
#include "lz_host.hpp"

#include <atomic>
#include <mutex>
#include <condition_variable>

#define SUBM(m, r, c) ((m)[((size_t)(r) << 8) | (size_t)(c)])

enum { OP_INS = 1, OP_DEL = 2, OP_SUB = 3 };
#define WORST_SCORE (-0x7FFFFFFF - 1)

// ---- reduce_to_points / segment_peak, src/gapped_extend.c:463-559
static u32 segment_peak(const u8* s1, const u8* s2, u32 len, const s32* sub)
{
    const u8 *t1 = s1, *t2 = s2;
    if (len <= 31) return len / 2;
    s32 sim = 0; u32 ix;
    for (ix = 0; ix < 31; ix++) sim += SUBM(sub, *t1++, *t2++);
    s32 best = sim; u32 peak = 31 / 2;
    for (; ix < len; ix++) {
        sim -= SUBM(sub, *s1++, *s2++);
        sim += SUBM(sub, *t1++, *t2++);
        if (sim > best) { best = sim; peak = ix - 31 / 2; }
    }
    return peak;
}

void lzh_reduce_to_points(const u8* t, const u8* q, const s32* sub, lz_segment* segs, u32 n)
{
    // independent per anchor: a few threads (8.8 -> ~2 ms for the 78 k anchors of a 50 Mbp strand)
    auto part = [&](u32 lo, u32 hi) {
        for (u32 i = lo; i < hi; i++) {
            u32 peak = segment_peak(t + segs[i].pos1, q + segs[i].pos2, segs[i].length, sub);
            segs[i].pos1 += peak; segs[i].pos2 += peak; segs[i].length = 0;
        }
    };
    const u32 nt = n < 8192 ? 1u : 8u;
    if (nt == 1) { part(0, n); return; }
    std::vector<std::thread> th;
    for (u32 k = 0; k < nt; k++) th.emplace_back(part, (u32)((u64)n * k / nt), (u32)((u64)n * (k + 1) / nt));
    for (auto& x : th) x.join();
}

// qSegmentsByDecreasingScore, src/segment.c:1748-1771 (a total order up to identical records)
static bool seg_before(const lz_segment& a, const lz_segment& b)
{
    if (a.s != b.s) return a.s > b.s;
    if (a.length != b.length) return a.length < b.length;
    if (a.pos2 != b.pos2) return a.pos2 < b.pos2;
    if (a.pos1 != b.pos1) return a.pos1 < b.pos1;
    return a.id < b.id;
}

struct Neighbours { s32 la, ls, ra, rs; };

// msp_left_right, src/gapped_extend.c:3953-4028.  rc: 1 = ok, 0 = anchor lies on an alignment, -1 = internal
// The reference walks every alignment that starts at or before pos1; only those that also end at or
// after pos1 matter.  With the running maximum of end1 along the start order (obi_maxend) the walk goes
// backwards from the last alignment starting <= pos1 and stops where nothing earlier can reach pos1;
// the survivors are then visited in the reference's (ascending) order, which decides ties.
static int msp_left_right(const LzHostSnapshot& S, u32 pos1, u32 pos2, Neighbours& nb)
{
    u32 right = 0xFFFFFFFFu, left = 0xFFFFFFFFu;
    nb.la = nb.ls = nb.ra = nb.rs = -1;
    size_t lo = 0, hi = S.obi.size();                          // first o with aligns[obi[o]].pos1 > pos1
    while (lo < hi) { const size_t m = (lo + hi) / 2; if (S.aligns[S.obi[m]].pos1 > pos1) hi = m; else lo = m + 1; }
    size_t cand[64]; size_t nc = 0; bool overflow = false;
    for (size_t o = hi; o-- > 0; ) {
        if (S.obi_maxend[o] < pos1) break;
        if (S.aligns[S.obi[o]].end1 < pos1) continue;
        if (nc == 64) { overflow = true; break; }
        cand[nc++] = o;
    }
    const size_t n_visit = overflow ? hi : nc;
    for (size_t v = 0; v < n_visit; v++) {
        const size_t o = overflow ? v : cand[nc - 1 - v];
        const LzDpAlign& al = S.aligns[S.obi[o]];
        if (al.end1 < pos1) continue;
        // the first piece that ends at or after pos1: e1 never decreases along an alignment (format_segments), so
        // the reference's linear walk over the pieces is a binary search (long alignments have hundreds of pieces)
        s32 slo = al.first_seg, shi = al.last_seg + 1;
        while (slo < shi) { const s32 m = slo + (shi - slo) / 2; if (S.segs[m].e1 >= pos1) shi = m; else slo = m + 1; }
        if (slo > al.last_seg) continue;
        const s32 bp = slo;
        const LzDpSeg& g = S.segs[bp];
        if (g.type == LZ_HORZ_SEG) return -1;
        s32 x = (g.type == LZ_DIAG_SEG) ? LZ_SDIFF(g.b2, pos2) + LZ_SDIFF(pos1, g.b1) : LZ_SDIFF(g.b2, pos2);
        if (x == 0) return 0;
        if (x > 0 && (u32)x < right) { right = (u32)x; nb.ra = S.obi[o]; nb.rs = bp; }
        else if (x < 0 && (u32)(-x) < left) { left = (u32)(-x); nb.la = S.obi[o]; nb.ls = bp; }
    }
    return 1;
}

// does (pos1, pos2) lie on alignment `al` -- msp_left_right's x == 0 for this one alignment
static bool on_alignment(const LzHostSnapshot& S, const LzDpAlign& al, u32 pos1, u32 pos2)
{
    if (al.pos1 > pos1 || al.end1 < pos1) return false;
    s32 slo = al.first_seg, shi = al.last_seg + 1;
    while (slo < shi) { const s32 m = slo + (shi - slo) / 2; if (S.segs[m].e1 >= pos1) shi = m; else slo = m + 1; }
    if (slo > al.last_seg) return false;
    const LzDpSeg& g = S.segs[slo];
    if (g.type == LZ_HORZ_SEG) return false;                    // (msp_left_right reports the internal error)
    const s32 x = (g.type == LZ_DIAG_SEG) ? LZ_SDIFF(g.b2, pos2) + LZ_SDIFF(pos1, g.b1) : LZ_SDIFF(g.b2, pos2);
    return x == 0;
}

// the same test against the pieces of an alignment that is not in the snapshot yet (what a commit would add)
static bool on_pieces(const std::vector<LzDpSeg>& segs, u32 pos1, u32 pos2)
{
    if (segs.empty() || segs.front().b1 > pos1 || segs.back().e1 < pos1) return false;
    size_t slo = 0, shi = segs.size();
    while (slo < shi) { const size_t m = slo + (shi - slo) / 2; if (segs[m].e1 >= pos1) shi = m; else slo = m + 1; }
    if (slo >= segs.size()) return false;
    const LzDpSeg& g = segs[slo];
    if (g.type == LZ_HORZ_SEG) return false;
    const s32 x = (g.type == LZ_DIAG_SEG) ? LZ_SDIFF(g.b2, pos2) + LZ_SDIFF(pos1, g.b1) : LZ_SDIFF(g.b2, pos2);
    return x == 0;
}

// the reference's walk (src/gapped_extend.c:3953-4028), kept as the yardstick of lzh_selftest_neighbours
static int msp_left_right_plain(const LzHostSnapshot& S, u32 pos1, u32 pos2, Neighbours& nb)
{
    u32 right = 0xFFFFFFFFu, left = 0xFFFFFFFFu;
    nb.la = nb.ls = nb.ra = nb.rs = -1;
    for (size_t o = 0; o < S.obi.size(); o++) {
        const LzDpAlign& al = S.aligns[S.obi[o]];
        if (al.pos1 > pos1) break;
        if (al.end1 < pos1) continue;
        s32 bp = -1;
        for (s32 k = al.first_seg; k <= al.last_seg; k++) if (S.segs[k].e1 >= pos1) { bp = k; break; }
        if (bp < 0) continue;
        const LzDpSeg& g = S.segs[bp];
        if (g.type == LZ_HORZ_SEG) return -1;
        s32 x = (g.type == LZ_DIAG_SEG) ? LZ_SDIFF(g.b2, pos2) + LZ_SDIFF(pos1, g.b1) : LZ_SDIFF(g.b2, pos2);
        if (x == 0) return 0;
        if (x > 0 && (u32)x < right) { right = (u32)x; nb.ra = S.obi[o]; nb.rs = bp; }
        else if (x < 0 && (u32)(-x) < left) { left = (u32)(-x); nb.la = S.obi[o]; nb.ls = bp; }
    }
    return 1;
}

// align_left_right, src/gapped_extend.c:4078-4180
struct LrAcc {
    u32 rob = 0xFFFFFFFFu, rot = 0xFFFFFFFFu, lob = 0xFFFFFFFFu, lot = 0xFFFFFFFFu;
    s32 m_rob = -1, m_rot = -1, m_lob = -1, m_lot = -1, b_rob = -1, b_rot = -1, b_lob = -1, b_lot = -1;
};
static void lr_visit(const LzHostSnapshot& S, s32 ai, u32 pos1, u32 pos2, u32 end1, u32 end2, LrAcc& A)
{
    const LzDpAlign& al = S.aligns[ai];
    s32 bp = -1, k; s32 x;
    for (k = al.first_seg; k <= al.last_seg; k++) if (S.segs[k].type != LZ_HORZ_SEG && S.segs[k].e1 >= pos1) { bp = k; break; }
    if (bp >= 0 && S.segs[bp].b1 <= pos1) {
        const LzDpSeg& g = S.segs[bp];
        x = (g.type == LZ_DIAG_SEG) ? LZ_SDIFF(g.b2, pos2) + LZ_SDIFF(pos1, g.b1) : LZ_SDIFF(g.b2, pos2);
        if (x > 0 && (u32)x < A.rob) { A.rob = (u32)x; A.m_rob = ai; A.b_rob = bp; }
        else if (x < 0 && (u32)(-x) < A.lob) { A.lob = (u32)(-x); A.m_lob = ai; A.b_lob = bp; }
    }
    if (bp >= 0) {
        s32 bq = -1;
        for (k = bp; k <= al.last_seg; k++) if (S.segs[k].type != LZ_HORZ_SEG && S.segs[k].e1 >= end1) { bq = k; break; }
        if (bq >= 0) {
            const LzDpSeg& g = S.segs[bq];
            x = (g.type == LZ_DIAG_SEG) ? LZ_SDIFF(g.b2, end2) + LZ_SDIFF(end1, g.b1) : LZ_SDIFF(g.b2, end2);
            if (x > 0 && (u32)x < A.rot) { A.rot = (u32)x; A.m_rot = ai; A.b_rot = bq; }
            else if (x < 0 && (u32)(-x) < A.lot) { A.lot = (u32)(-x); A.m_lot = ai; A.b_lot = bq; }
        }
    }
}
static void lr_store(LzDpAlign& m, const LrAcc& A)
{
    m.right_align1 = A.m_rob; m.right_seg1 = A.b_rob; m.right_align2 = A.m_rot; m.right_seg2 = A.b_rot;
    m.left_align1 = A.m_lob;  m.left_seg1 = A.b_lob;  m.left_align2 = A.m_lot;  m.left_seg2 = A.b_lot;
}
static void align_left_right(const LzHostSnapshot& S, LzDpAlign& m)
{
    const u32 pos1 = m.pos1, pos2 = m.pos2, end1 = m.end1, end2 = m.end2;
    LrAcc A;
    // The reference walks every alignment and skips those that do not overlap [pos1, end1] in the target.  obi is ordered by
    // pos1 and obi_maxend[o] is the largest end1 of obi[0..o]: everything before the first o whose running maximum
    // reaches pos1 ends before pos1, everything from the first alignment that starts after end1 on starts after it --
    // the same alignments in the same order, without the quadratic walk (9 k alignments at the north star's size).
    size_t o_lo = 0;
    { size_t lo = 0, hi = S.obi_maxend.size(); while (lo < hi) { const size_t mid = lo + (hi - lo) / 2; if (S.obi_maxend[mid] >= pos1) hi = mid; else lo = mid + 1; } o_lo = lo; }
    for (size_t o = o_lo; o < S.obi.size(); o++) {
        const LzDpAlign& al = S.aligns[S.obi[o]];
        if (al.pos1 > end1) break;
        if (al.end1 < pos1) continue;
        lr_visit(S, S.obi[o], pos1, pos2, end1, end2, A);
    }
    lr_store(m, A);
}
// the reference's walk over every alignment, the yardstick of lzh_selftest_neighbours
static void align_left_right_plain(const LzHostSnapshot& S, LzDpAlign& m)
{
    LrAcc A;
    for (size_t o = 0; o < S.obi.size(); o++) {
        const LzDpAlign& al = S.aligns[S.obi[o]];
        if (al.pos1 > m.end1 || al.end1 < m.pos1) continue;
        lr_visit(S, S.obi[o], m.pos1, m.pos2, m.end1, m.end2, A);
    }
    lr_store(m, A);
}

// get_above_below, src/gapped_extend.c:4043-4059: the first entry of oed that ends before a1 / of obi that starts after it
// (the reference's linear walks as bisections: oed is ordered by decreasing end, obi by increasing start)
static void above_below(const LzHostSnapshot& S, u32 a1, s32& below, s32& above)
{
    below = -1; above = -1;
    { size_t lo = 0, hi = S.oed.size(); while (lo < hi) { const size_t mid = lo + (hi - lo) / 2; if (S.aligns[S.oed[mid]].end1 >= a1) lo = mid + 1; else hi = mid; } if (lo < S.oed.size()) below = (s32)lo; }
    { size_t lo = 0, hi = S.obi.size(); while (lo < hi) { const size_t mid = lo + (hi - lo) / 2; if (S.aligns[S.obi[mid]].pos1 <= a1) lo = mid + 1; else hi = mid; } if (lo < S.obi.size()) above = (s32)lo; }
}
static void above_below_plain(const LzHostSnapshot& S, u32 a1, s32& below, s32& above)
{
    below = -1; above = -1;
    for (size_t o = 0; o < S.oed.size(); o++) if (S.aligns[S.oed[o]].end1 < a1) { below = (s32)o; break; }
    for (size_t o = 0; o < S.obi.size(); o++) if (S.aligns[S.obi[o]].pos1 > a1) { above = (s32)o; break; }
}

// insert_align, src/gapped_extend.c:4210-4245
static void insert_align(LzHostSnapshot& S, s32 ai)
{
    const LzDpAlign& m = S.aligns[ai];
    // (the reference's linear walks to the first entry with pos1 >= m.pos1 / end1 <= m.end1, as binary searches)
    size_t p;
    { size_t lo = 0, hi = S.obi.size(); while (lo < hi) { const size_t mid = lo + (hi - lo) / 2; if (S.aligns[S.obi[mid]].pos1 < m.pos1) lo = mid + 1; else hi = mid; } p = lo; }
    S.obi.insert(S.obi.begin() + p, ai);
    const size_t p_obi = p;
    { size_t lo = 0, hi = S.oed.size(); while (lo < hi) { const size_t mid = lo + (hi - lo) / 2; if (S.aligns[S.oed[mid]].end1 > m.end1) lo = mid + 1; else hi = mid; } p = lo; }
    S.oed.insert(S.oed.begin() + p, ai);
    // the running maximum of end1 along obi: unchanged in front of the new entry; behind it every value is the old one or
    // the new alignment's end, whichever is larger -- and the old values only grow, so the walk stops at the first that
    // is large enough (it used to recompute the whole tail: quadratic in the alignments)
    const u32 before = p_obi ? S.obi_maxend[p_obi - 1] : 0u;
    S.obi_maxend.insert(S.obi_maxend.begin() + p_obi, before > m.end1 ? before : m.end1);
    for (size_t o = p_obi + 1; o < S.obi_maxend.size() && S.obi_maxend[o] < m.end1; o++) S.obi_maxend[o] = m.end1;
}

// score_alignment, src/gapped_extend.c:5631-5675
static s32 score_alignment(const LzGappedParams& G, u32 pos1, u32 pos2, const std::vector<u32>& sc)
{
    const u8 *s1 = G.t + pos1, *s2 = G.q + pos2;
    s32 sim = 0;
    for (u32 w : sc) {
        u32 rpt = w >> 2, op = w & 3;
        if (rpt == 0) continue;
        if (op == OP_SUB) { const u8* stop = s1 + rpt; while (s1 < stop) sim += SUBM(G.sub, *(s1++), *(s2++)); }
        else if (op == OP_INS) { sim -= G.gap_open + (s32)(rpt * (u32)G.gap_extend); s2 += rpt; }
        else if (op == OP_DEL) { sim -= G.gap_open + (s32)(rpt * (u32)G.gap_extend); s1 += rpt; }
    }
    return sim;
}

struct Built {                          // ydrop_align's outputs (alignio)
    s32 s; u32 start1, start2, stop1, stop2;
    std::vector<u32> script;
};

// ydrop_align after the two one-sided DPs, src/gapped_extend.c:2520-2583
static void splice_and_trim(const LzGappedParams& G, u32 a1, u32 a2, const LzDpResult& rl, const std::vector<u32>& ol,
                            const LzDpResult& rr, const std::vector<u32>& orr, Built& b)
{
    b.start1 = a1 + 1 - rl.end1; b.start2 = a2 + 1 - rl.end2;
    b.stop1 = a1 + rr.end1;      b.stop2 = a2 + rr.end2;
    b.script = ol;                                           // left half: traceback order == forward order
    if (!orr.empty()) {                                      // edit_script_reverse + edit_script_append
        size_t k = orr.size();
        u32 first = orr[k - 1];
        if (!b.script.empty() && (b.script.back() & 3) == (first & 3)) { b.script.back() += (first >> 2) << 2; k--; }
        while (k > 0) b.script.push_back(orr[--k]);
    }
    b.s = rr.score + rl.score;
    std::vector<u32>& sc = b.script;
    if (sc.empty()) return;
    if ((sc[0] & 3) != OP_SUB) {                             // lop_initial_indels, :2589-2635
        u32 p1 = b.start1, p2 = b.start2; size_t k;
        for (k = 0; k < sc.size(); k++) {
            u32 op = sc[k] & 3, rpt = sc[k] >> 2;
            if (op == OP_SUB) break; else if (op == OP_INS) p2 += rpt; else if (op == OP_DEL) p1 += rpt;
        }
        if (k == sc.size()) b.s = WORST_SCORE;
        else {
            b.start1 = p1; b.start2 = p2;
            sc.erase(sc.begin(), sc.begin() + k);
            b.s = score_alignment(G, b.start1, b.start2, sc);
        }
    }
    if ((sc.back() & 3) != OP_SUB) {                         // lop_final_indels, :2640-2683
        u32 p1 = b.stop1, p2 = b.stop2; size_t k;
        for (k = sc.size(); k > 0;) {
            k--;
            u32 op = sc[k] & 3, rpt = sc[k] >> 2;
            if (op == OP_SUB) { k++; break; } else if (op == OP_INS) p2 -= rpt; else if (op == OP_DEL) p1 -= rpt;
        }
        if (k == 0) b.s = WORST_SCORE;
        else {
            b.stop1 = p1; b.stop2 = p2;
            sc.resize(k);
            b.s = score_alignment(G, b.start1, b.start2, sc);
        }
    }
}

// format_alignment + save_seg, src/gapped_extend.c:5153-5275: script -> diag / horz / vert pieces
static void format_segments(const Built& b, std::vector<LzDpSeg>& segs)
{
    segs.clear();
    const u32 beg1 = b.start1 + 1, end1 = b.stop1 + 1, beg2 = b.start2 + 1, end2 = b.stop2 + 1;
    const u32 height = end1 - beg1 + 1, width = end2 - beg2 + 1;
    u32 i = 0, j = 0; size_t k = 0;
    while (i < height || j < width) {
        u32 si = i, sj = j, run = 0;
        while (k < b.script.size() && (b.script[k] & 3) == OP_SUB) { run += b.script[k] >> 2; k++; }
        i += run; j += run;
        LzDpSeg d; d.type = LZ_DIAG_SEG; d.b1 = beg1 + si - 1; d.b2 = beg2 + sj - 1; d.e1 = beg1 + i - 2; d.e2 = beg2 + j - 2;
        if (!segs.empty()) {
            const LzDpSeg& tail = segs.back();
            LzDpSeg c; c.type = (d.b1 == tail.e1 + 1) ? LZ_HORZ_SEG : LZ_VERT_SEG;
            c.b1 = tail.e1 + 1; c.b2 = tail.e2 + 1; c.e1 = d.b1 - 1; c.e2 = d.b2 - 1;
            segs.push_back(c);
        }
        segs.push_back(d);
        if (i < height || j < width) {
            if (k < b.script.size()) {
                u32 op = b.script[k] & 3, rpt = b.script[k] >> 2;
                if (op == OP_INS) j += rpt; else if (op == OP_DEL) i += rpt;
                k++;
            }
        }
    }
}

// does alignment `al` have a cell inside the closed rectangle [r0,r1] x [c0,c1] (target x query)?
static bool align_touches(const LzHostSnapshot& S, const LzDpAlign& al, s64 r0, s64 r1, s64 c0, s64 c1)
{
    if ((s64)al.pos1 > r1 || (s64)al.end1 < r0) return false;
    for (s32 k = al.first_seg; k <= al.last_seg; k++) {
        const LzDpSeg& g = S.segs[k];
        if ((s64)g.b1 > r1 || (s64)g.e1 < r0 || (s64)g.b2 > c1 || (s64)g.e2 < c0) {
            if (g.type != LZ_HORZ_SEG && g.type != LZ_VERT_SEG) continue;
            if ((s64)g.b1 > r1 + 1 || (s64)g.e1 + 1 < r0 || (s64)g.b2 > c1 + 1 || (s64)g.e2 + 1 < c0) continue;
        }
        if (g.type == LZ_DIAG_SEG) {
            s64 lo = std::max<s64>(std::max<s64>(r0 - (s64)g.b1, c0 - (s64)g.b2), 0);
            s64 hi = std::min<s64>(std::min<s64>(r1 - (s64)g.b1, c1 - (s64)g.b2), (s64)g.e1 - (s64)g.b1);
            if (lo <= hi) return true;
        } else return true;                                    // gap pieces: bounding box overlap is enough
    }
    return false;
}

// does an alignment committed since the snapshot (index >= k0) touch one of the two rectangles a cached DP pair explored?
struct Rect2 { s64 lr0, lr1, lc0, lc1, rr0, rr1, rc0, rc1; };
static bool touched_since(const LzHostSnapshot& S, size_t k0, const Rect2& q)
{
    const s64 row_lo = q.lr0 < q.rr0 ? q.lr0 : q.rr0, row_hi = q.lr1 > q.rr1 ? q.lr1 : q.rr1;
    size_t lo = 0, hi = S.obi_maxend.size();
    while (lo < hi) { const size_t mid = lo + (hi - lo) / 2; if ((s64)S.obi_maxend[mid] >= row_lo) hi = mid; else lo = mid + 1; }
    for (size_t o = lo; o < S.obi.size(); o++) {
        const s32 ai = S.obi[o];
        const LzDpAlign& al = S.aligns[ai];
        if ((s64)al.pos1 > row_hi) break;
        if ((size_t)ai < k0) continue;
        if (align_touches(S, al, q.lr0, q.lr1, q.lc0, q.lc1) || align_touches(S, al, q.rr0, q.rr1, q.rc0, q.rc1)) return true;
    }
    return false;
}
static bool touched_since_plain(const LzHostSnapshot& S, size_t k0, const Rect2& q)       // (every alignment: the yardstick)
{
    for (size_t ai = k0; ai < S.aligns.size(); ai++)
        if (align_touches(S, S.aligns[ai], q.lr0, q.lr1, q.lc0, q.lc1) || align_touches(S, S.aligns[ai], q.rr0, q.rr1, q.rc0, q.rc1)) return true;
    return false;
}

// lookup_partition (src/sequences.c:6536-...): the limits [low, high) of the partition holding pos
static bool partition_limits(const u32* sep, u32 n_sep, u32 pos, u32 len, u32& low, u32& high)
{
    if (!sep) { low = 0; high = len; return true; }
    const u32* hi = std::upper_bound(sep, sep + n_sep, pos);          // first separator beyond pos
    if (hi == sep || hi == sep + n_sep || hi[-1] == pos) return false;  // outside every partition / on a separator
    low = hi[-1] + 1; high = hi[0];
    return true;
}

static bool same_bases(const u8* a, const u8* b, u32 n)
{
    for (u32 i = 0; i < n; i++) {
        u8 x = a[i], y = b[i];
        if (x >= 'a' && x <= 'z') x -= 32;
        if (y >= 'a' && y <= 'z') y -= 32;
        if (x != y) return false;
    }
    return true;
}

int lzh_gapped_extend(const LzGappedParams& G, LzDpExecutor& exec, lz_segment* anchors, u32 n_anchors,
                      std::vector<lz_align>& out, std::vector<u32>& out_ops, LzGappedStats& st)
{
    out.clear(); out_ops.clear();
    memset(&st, 0, sizeof(st));
    // The trivial alignments the reference puts in front of the anchors (src/gapped_extend.c:1118-1290):
    //  * neither sequence partitioned: the self-alignment when identical_sequences says so (:1886-1933: dna_toupper() of
    //    the bytes equal, same strand flags);
    //  * seq1 partitioned, seq2 not: ONE, for the first partition of seq1 that is seq2 (identical_partition_of_sequence,
    //    :2034-2113);
    //  * both partitioned: one per partition pair (k, k) when ALL pairs are identical (identical_partitioned_sequences,
    //    :1952-1997).
    // With inhibitTrivial and partitions but no such "partitioned triviality" the reference instead drops, at output time,
    // alignments that cover a whole partition pair of equal NAME (delayedCheckForTrivial, :1485-1545): names do not cross
    // this ABI, so a result that holds a candidate for that test is declined at the end (below).
    struct Triv { u32 lo1, lo2, len; };
    std::vector<Triv> triv;
    bool delayed_check = false;
    auto n_parts = [](const u32* sep, u32 n_sep) { return sep ? (n_sep ? n_sep - 1 : 0u) : 1u; };
    auto part_of = [&](const u32* sep, u32 k, u32 whole, u32& lo, u32& hi) { if (sep) { lo = sep[k] + 1; hi = sep[k + 1]; } else { lo = 0; hi = whole; } };
    if (G.sep1) for (u32 k = 0; k + 1 < G.n_sep1; k++) if (G.sep1[k + 1] > G.tlen || G.sep1[k] + 1 > G.sep1[k + 1]) return LZGPU_ERR_ARG;
    if (G.sep2) for (u32 k = 0; k + 1 < G.n_sep2; k++) if (G.sep2[k + 1] > G.qlen || G.sep2[k] + 1 > G.sep2[k + 1]) return LZGPU_ERR_ARG;
    if (!G.sep1 && !G.sep2) {
        if (!G.strands_differ && G.tlen == G.qlen && G.tlen > 0 && same_bases(G.t, G.q, G.tlen)) triv.push_back({ 0u, 0u, G.tlen });
    } else if (G.sep1 && !G.sep2) {
        bool found = false;
        if (!G.strands_differ)
            for (u32 k = 0; k < n_parts(G.sep1, G.n_sep1) && !found; k++) {
                u32 lo, hi; part_of(G.sep1, k, G.tlen, lo, hi);
                if (hi - lo != G.qlen) continue;
                if (G.qlen == 0) return LZGPU_NH_IDENTICAL;               // (an empty sequence "identical" to an empty partition: left to the reference)
                if (same_bases(G.t + lo, G.q, G.qlen)) { triv.push_back({ lo, 0u, G.qlen }); found = true; }
            }
        delayed_check = G.inhibit_trivial && !found;
    } else if (G.sep1 && G.sep2) {
        bool all = !G.strands_differ && n_parts(G.sep1, G.n_sep1) == n_parts(G.sep2, G.n_sep2);
        for (u32 k = 0; all && k < n_parts(G.sep1, G.n_sep1); k++) {
            u32 lo1, hi1, lo2, hi2; part_of(G.sep1, k, G.tlen, lo1, hi1); part_of(G.sep2, k, G.qlen, lo2, hi2);
            if (hi1 - lo1 != hi2 - lo2 || !same_bases(G.t + lo1, G.q + lo2, hi1 - lo1)) all = false;
        }
        if (all)
            for (u32 k = 0; k < n_parts(G.sep1, G.n_sep1); k++) {
                u32 lo1, hi1, lo2, hi2; part_of(G.sep1, k, G.tlen, lo1, hi1); part_of(G.sep2, k, G.qlen, lo2, hi2);
                if (hi1 == lo1) return LZGPU_NH_IDENTICAL;                // (empty partitions: left to the reference)
                triv.push_back({ lo1, lo2, hi1 - lo1 });
            }
        delayed_check = G.inhibit_trivial && !all;
    } else delayed_check = G.inhibit_trivial;                             // (seq2 alone partitioned: no partitioned triviality, :1124-1147)
    if (G.gap_extend <= 0) return LZGPU_NH_UNSUPPORTED;

    // LZGPU_HOSTPROF=1: where the host time of the stage goes
    static const bool prof = getenv("LZGPU_HOSTPROF") != nullptr;
    double t_sort = 0, t_window = 0, t_exec = 0, t_prebuild = 0, t_commit = 0, t_c_lr = 0, t_c_chk = 0, t_c_build = 0, t_c_alr = 0, t_c_ins = 0, t_c_cov = 0;
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_mark = now();
    auto lap = [&](double& acc) { const double t = now(); acc += t - t_mark; t_mark = t; };
    lzh_sort4(anchors, anchors + n_anchors, seg_before);       // batched_segments, :1675
    st.anchors = n_anchors;
    lap(t_sort);

    LzHostSnapshot S;
    struct Info { s32 s; u32 beg1, beg2, end1, end2; std::vector<u32> script; bool trivial = false; };
    std::vector<Info> info;                                    // parallel to S.aligns
    for (const Triv& tv : triv) {
        // a trivial alignment bounds every anchor from the start, :1152-1290 (one diagonal segment; its score saturates
        // at bestPossibleScore and is raised to the threshold "so it won't be discarded")
        s32 sc = 0;
        for (u32 i = 0; i < tv.len; i++) {
            u8 a = G.t[tv.lo1 + i], b = G.q[tv.lo2 + i];
            if (a >= 'a' && a <= 'z') a -= 32;
            if (b >= 'a' && b <= 'z') b -= 32;
            const s32 w = G.sub[(u32)a * 256 + b];
            if (sc == 0x7FFFFFFF) ;
            else if (w <= 0 || sc < 0x7FFFFFFF - w) sc += w;
            else sc = 0x7FFFFFFF;
        }
        LzDpAlign m; memset(&m, 0, sizeof(m));
        m.pos1 = tv.lo1; m.pos2 = tv.lo2; m.end1 = tv.lo1 + tv.len - 1; m.end2 = tv.lo2 + tv.len - 1;
        m.first_seg = m.last_seg = (s32)S.segs.size();
        m.left_align1 = m.right_align1 = m.left_align2 = m.right_align2 = -1;
        m.left_seg1 = m.right_seg1 = m.left_seg2 = m.right_seg2 = -1;
        LzDpSeg g; g.b1 = m.pos1; g.b2 = m.pos2; g.e1 = m.end1; g.e2 = m.end2; g.type = LZ_DIAG_SEG;
        S.segs.push_back(g); S.aligns.push_back(m);
        Info in; in.s = sc < G.score_thresh ? G.score_thresh : sc; in.beg1 = tv.lo1 + 1; in.beg2 = tv.lo2 + 1; in.end1 = tv.lo1 + tv.len; in.end2 = tv.lo2 + tv.len; in.trivial = true;
        for (u32 left = tv.len; left; ) { const u32 n = left < 0x3FFFFFFFu ? left : 0x3FFFFFFFu; in.script.push_back((n << 2) | 3u); left -= n; }   // edit_script_sub
        info.push_back(std::move(in));
        insert_align(S, (s32)S.aligns.size() - 1);
    }

    const u32 W = G.window ? G.window : 1024;
    u64 paired_bases = 0;
    static const u32 helper_min = []() { const char* e = getenv("LZGPU_HELPER_MIN_ANCHORS"); return (u32)(e ? atoi(e) : 20000); }();   // (tests: 0)
    // (three helpers for a 50 Mbp strand's 79 k anchors; seven from 150 k anchors on -- a 200 Mbp strand has 300 k, 9 k alignments to build per round:
    // gapped stage 0.230-0.245 -> 0.222-0.231 s; LZGPU_HELPERS fixes the number)
    static const u32 helper_env = []() { const char* e = getenv("LZGPU_HELPERS"); const int v = e ? atoi(e) : 0; return (u32)(v > 0 ? v : 0); }();
    const u32 helper_n = helper_env ? helper_env : (n_anchors >= 150000u ? 7u : 3u);
    ForkJoin helpers(n_anchors >= helper_min ? helper_n : 0u); // (small problems -- tweener windows -- stay on their own thread)
    // Which anchors of a window are worth a speculative DP.  Most anchors lie on the alignment an
    // earlier (better) anchor is about to produce -- the reference drops them in msp_left_right
    // without running a DP (98.5 % on the 10 Mbp pair).  An anchor within
    // NEAR_DIAG diagonals and NEAR_POS target bases of an anchor already selected in this window
    // is therefore deferred: when the commit pass reaches it, it is either on a committed
    // alignment (skipped, as in the reference) or it starts the next window.  This only steers
    // what is speculated; what is committed is decided by the checks below.
    const s64 NEAR_DIAG = 1500, NEAR_POS = 60000, TIGHT_DIAG = 300;
    const u32 INSURE = 256;
    u32 next = 0;
    std::vector<LzDpJob> jobs; std::vector<LzDpResult> res; std::vector<std::vector<u32>> ops;
    struct Entry { u32 anchor_ix; bool speculated; u32 near_slot; };   // near_slot: the selected anchor a deferred one was found near (its ext_l / ext_r slot)
    // A finished speculative DP pair stays usable across windows for as long as its validity
    // conditions hold against the alignments committed after the snapshot it ran against.
    struct Cached { u32 a1, a2; Neighbours nb; size_t n_snap; LzDpResult rl, rr; std::vector<u32> ol, orr; };
    // (indexed by anchor: the window scan looks every anchor up -- 3 x 10^5 per strand of the 200 Mbp pair, nearly all of them
    // misses -- and a hash table made that look-up the scan's whole cost: 15 of its 16 ms)
    struct AnchorCache {
        std::vector<s32> ix; std::deque<Cached> pool; std::vector<s32> free_list;
        explicit AnchorCache(size_t n) : ix(n, -1) {}
        Cached* find(u32 j) { return ix[j] >= 0 ? &pool[(size_t)ix[j]] : nullptr; }
        void erase(u32 j) { if (ix[j] >= 0) { pool[(size_t)ix[j]] = Cached(); free_list.push_back(ix[j]); ix[j] = -1; } }
        Cached& emplace(u32 j, Cached&& c)
        {
            s32 k;
            if (!free_list.empty()) { k = free_list.back(); free_list.pop_back(); pool[(size_t)k] = std::move(c); }
            else { k = (s32)pool.size(); pool.push_back(std::move(c)); }
            ix[j] = k; return pool[(size_t)k];
        }
    } cache(n_anchors);
    struct Prebuilt { Built b; std::vector<LzDpSeg> segs; bool have = false; };
    std::vector<Prebuilt> prebuilt;                            // per SPECULATED entry of the window: what its commit would build
    std::vector<u32> spec_of, spec_list;                       // entry -> its place in prebuilt / the speculated entries
    std::vector<Entry> entries;
    std::vector<u32> fresh;                                    // anchors launched in this round
    // the alignment committed for the selected anchor a deferred anchor was found near, remembered across windows:
    // when a window is cut, the anchors behind the cut are scanned again and most of them lie on that alignment
    std::vector<s32> near_align(n_anchors, -1);
    struct Chosen { s64 dg, a1; u32 ext; };                    // a selected anchor of this window, and its slot in ext_l / ext_r
    // cells of NEAR_DIAG diagonals, directly indexed (diagonals run from -qlen to tlen); `touched` lists the cells in use
    const s64 cell_lo = -(s64)(G.qlen / NEAR_DIAG) - 2;
    std::vector<std::vector<Chosen>> chosen_grid((size_t)((s64)(G.tlen / NEAR_DIAG) + 2 - cell_lo + 1));
    std::vector<u32> touched;
    // How far the deferred anchors near a selected one reach on either side of it: a guess at the rows its two DPs
    // will sweep (the anchors an alignment swallows line up along it).  Only the launch order uses it -- a launch
    // lasts as long as its longest DP, so the long ones should be among the first resident.
    std::vector<u32> ext_l, ext_r;
    struct Member { u32 e, pos1, pos2; };                      // a deferred entry (index into entries) and its anchor
    std::vector<std::vector<Member>> slot_members;             // the deferred entries found near a slot's selected anchor
    std::vector<u8> covered;                                   // commit pass: the deferred entry lies on its slot's alignment (checked when that was committed)
    std::unordered_map<u32, u32> ext_of;                       // anchor index -> slot
    std::vector<std::pair<s64, s64>> slot_anchor;              // LZGPU_HOSTPROF: (diagonal, pos1) of a slot's selected anchor
    while (next < n_anchors) {
        // ---- speculation window against the current snapshot
        jobs.clear(); entries.clear(); fresh.clear(); ext_l.clear(); ext_r.clear(); ext_of.clear(); slot_anchor.clear();
        for (auto& v : slot_members) v.clear();
        for (u32 t : touched) chosen_grid[t].clear();
        touched.clear();
        u32 insured = 0;
        u32 j = next;
        const u32 scan_limit = 64 * W;
        const size_t n_snap = S.aligns.size();
        for (; j < n_anchors && fresh.size() < W && entries.size() < scan_limit; j++) {
            const u32 a1 = anchors[j].pos1, a2 = anchors[j].pos2;
            Neighbours nb;
            if (near_align[j] >= 0 && on_alignment(S, S.aligns[near_align[j]], a1, a2)) { cache.erase(j); continue; }
            int ok = msp_left_right(S, a1, a2, nb);
            if (ok < 0) return LZGPU_ERR_STATE;
            if (ok == 0) { cache.erase(j); continue; }         // on an earlier alignment: gone for good
            const s64 dg = (s64)a1 - (s64)a2;
            Cached* const hit = cache.find(j);
            // (the selected anchors are kept in cells of NEAR_DIAG diagonals: a near one is in the anchor's
            // cell or one next to it -- a window scans up to 64 K anchors against up to 1 K selected ones)
            const s64 cell = (dg >= 0 ? dg : dg - (NEAR_DIAG - 1)) / NEAR_DIAG;
            if (hit == nullptr) {
                // 0 = not near, 1 = near (loose), 2 = near and almost on the same diagonal (tight)
                int near = 0; u32 near_slot = 0;
                for (s64 cc = cell - 1; cc <= cell + 1 && near < 2; cc++) {
                    const s64 gi = cc - cell_lo;
                    if (gi < 0 || gi >= (s64)chosen_grid.size()) continue;
                    for (auto& c : chosen_grid[(size_t)gi])
                        if (c.dg - dg <= NEAR_DIAG && dg - c.dg <= NEAR_DIAG &&
                            c.a1 - (s64)a1 <= NEAR_POS && (s64)a1 - c.a1 <= NEAR_POS) {
                            near = (c.dg - dg <= TIGHT_DIAG && dg - c.dg <= TIGHT_DIAG) ? 2 : (near < 1 ? 1 : near);
                            near_slot = c.ext;
                            if ((s64)a1 < c.a1) { const u32 d = (u32)(c.a1 - (s64)a1); if (d > ext_l[c.ext]) ext_l[c.ext] = d; }
                            else                { const u32 d = (u32)((s64)a1 - c.a1); if (d > ext_r[c.ext]) ext_r[c.ext] = d; }
                            if (near == 2) break;
                        }
                }
                // a loosely near anchor is usually on the selected anchor's alignment too, but when it is not it
                // costs a whole extra round for one DP: a bounded number of them is speculated anyway
                if (near == 1 && insured < INSURE) { insured++; near = 0; }
                if (near) { slot_members[near_slot].push_back({ (u32)entries.size(), a1, a2 }); entries.push_back({ j, false, near_slot }); continue; }
            }
            ext_of[j] = (u32)ext_l.size();
            { const size_t gi = (size_t)(cell - cell_lo); if (chosen_grid[gi].empty()) touched.push_back((u32)gi); chosen_grid[gi].push_back({ dg, (s64)a1, (u32)ext_l.size() }); }
            ext_l.push_back(0); ext_r.push_back(0);
            if (slot_members.size() < ext_l.size()) slot_members.emplace_back();
            if (prof) slot_anchor.push_back({ dg, (s64)a1 });
            entries.push_back({ j, true, (u32)ext_l.size() - 1 });
            if (hit != nullptr) continue;                      // result of an earlier round, re-validated at commit
            // get_above_below, :4043-4059
            s32 below, above;
            above_below(S, a1, below, above);
            // the partition holding the anchor bounds its extension, :1356-1372 / ydrop_align :2515-2531
            u32 low1, high1, low2, high2;
            if (!partition_limits(G.sep1, G.n_sep1, a1, G.tlen, low1, high1) || !partition_limits(G.sep2, G.n_sep2, a2, G.qlen, low2, high2)
                || a1 + 1 > high1 || a2 + 1 > high2) return LZGPU_ERR_STATE;
            LzDpJob L; memset(&L, 0, sizeof(L));
            L.anchor1 = a1; L.anchor2 = a2; L.reversed = 1; L.M = a1 + 1 - low1; L.N = a2 + 1 - low2;
            L.left_align = nb.la; L.left_seg = nb.ls; L.right_align = nb.ra; L.right_seg = nb.rs;
            L.list_start = below;
            LzDpJob R = L;
            R.reversed = 0; R.M = high1 - (a1 + 1); R.N = high2 - (a2 + 1); R.list_start = above;
            jobs.push_back(L); jobs.push_back(R);
            Cached cr; cr.a1 = a1; cr.a2 = a2; cr.nb = nb; cr.n_snap = n_snap;
            cache.emplace(j, std::move(cr));
            fresh.push_back(j);
        }
        lap(t_window);
        if (prof && st.rounds >= 1) { fprintf(stderr, "[lzgpu hostprof] round %u speculates anchors (rank:score):", (unsigned)st.rounds + 1); for (size_t k = 0; k < fresh.size() && k < 12; k++) fprintf(stderr, " %u:%d", fresh[k], anchors[fresh[k]].s); fprintf(stderr, " of %u entries\n", (unsigned)entries.size()); }
        if (entries.empty()) { next = j; break; }
        for (size_t k = 0; k < fresh.size(); k++) {            // (the extents kept growing while the window was scanned)
            const u32 e = ext_of[fresh[k]];
            jobs[2 * k].est_rows = ext_l[e]; jobs[2 * k + 1].est_rows = ext_r[e];
        }
        if (!jobs.empty()) {
            res.assign(jobs.size(), LzDpResult());
            ops.assign(jobs.size(), std::vector<u32>());
            int rc = exec.run(S, jobs, res, ops);
            if (rc) return rc;
            for (size_t k = 0; k < fresh.size(); k++) {
                Cached& cr = *cache.find(fresh[k]);
                cr.rl = res[2 * k]; cr.rr = res[2 * k + 1];
                cr.ol.swap(ops[2 * k]); cr.orr.swap(ops[2 * k + 1]);
            }
            for (const LzDpResult& r : res) st.dp_rows += r.max_row;
        }
        st.rounds++; st.dp_runs += jobs.size();
        lap(t_exec);
        // ---- what a commit builds from a DP pair -- the spliced script, its end trimming and rescoring, the pieces -- depends on
        // the pair alone: done for every speculated entry of the window up front, on the helpers (splice_and_trim walks both
        // sequences along the whole alignment: 2/3 of the serial pass it is taken out of)
        spec_of.assign(entries.size(), 0xFFFFFFFFu);
        spec_list.clear();
        for (size_t e = 0; e < entries.size(); e++) if (entries[e].speculated) { spec_of[e] = (u32)spec_list.size(); spec_list.push_back((u32)e); }
        prebuilt.clear(); prebuilt.resize(spec_list.size());
        covered.assign(entries.size(), 0);
        {
            const std::function<void(size_t, size_t)> build = [&](size_t lo, size_t hi) {
                for (size_t k = lo; k < hi; k++) {
                    const Cached* it = cache.find(entries[spec_list[k]].anchor_ix);   // (concurrent look-ups only: nothing is inserted or erased here)
                    if (it == nullptr) continue;
                    const Cached& sp = *it;
                    Prebuilt& pb = prebuilt[k];
                    splice_and_trim(G, sp.a1, sp.a2, sp.rl, sp.ol, sp.rr, sp.orr, pb.b);
                    format_segments(pb.b, pb.segs);
                    pb.have = true;
                    // ... and which of the deferred anchors found near this one lie on that alignment (a slot's members belong
                    // to this entry alone: no two threads write the same flag); it counts once the alignment is committed
                    const u32 slot = entries[spec_list[k]].near_slot;
                    if (slot < slot_members.size()) for (const Member& mb : slot_members[slot]) if (on_pieces(pb.segs, mb.pos1, mb.pos2)) covered[mb.e] = 1;
                }
            };
            helpers.begin_burst();
            helpers.run(spec_list.size(), helper_min ? 16 : 1, build);
            helpers.end_burst();
        }
        lap(t_prebuild);

        // ---- commit in the reference's order
        bool cut = false;
        std::vector<s32> slot_align(ext_l.size(), -1);        // alignment committed for a selected anchor of this window
        for (size_t e = 0; e < entries.size(); e++) {
            const u32 aix = entries[e].anchor_ix;
            Neighbours nb;
            // A deferred anchor nearly always lies on the alignment of the selected anchor it was found near, and "on an
            // alignment" needs no more than one witness (:3953-4028): that was tested for all of a slot's deferred
            // anchors at once when the slot's alignment was committed (below); one flag to read here.  (A deferred
            // anchor has no cached DP: nothing to erase.)
            if (covered[e] && !entries[e].speculated && entries[e].near_slot < slot_align.size() && slot_align[entries[e].near_slot] >= 0) continue;
            const double tq0 = prof ? now() : 0;
            int ok = msp_left_right(S, anchors[aix].pos1, anchors[aix].pos2, nb);
            if (prof) t_c_lr += now() - tq0;
            if (ok < 0) return LZGPU_ERR_STATE;
            if (ok == 0) { cache.erase(aix); continue; }       // lies on an alignment committed meanwhile
            if (!entries[e].speculated) {                       // needs a DP: head of the next window
                // (On the bench pair every strand has one or two of these: an anchor a few diagonals beside the long
                // alignment of the selected anchor it was found near, 8 kbp along it.  Its DP is bounded by that very
                // alignment, so it cannot be launched before the alignment exists: the second round is inherent.  A rule
                // that also speculated the first anchor beyond every 1.5 kbp gap in the run of deferred anchors found
                // nothing to add here and cost 5 ms of sorting -- not kept.)
                if (prof && entries[e].near_slot < slot_anchor.size())
                    fprintf(stderr, "[lzgpu hostprof] window cut at anchor %u (score %d): deferred near a selected anchor %lld diagonals and %lld bases away, not on its alignment\n",
                            aix, anchors[aix].s, (long long)((s64)anchors[aix].pos1 - (s64)anchors[aix].pos2 - slot_anchor[entries[e].near_slot].first),
                            (long long)((s64)anchors[aix].pos1 - slot_anchor[entries[e].near_slot].second));
                next = aix; cut = true; break;
            }
            const Cached& sp = *cache.find(aix);
            const LzDpResult& rl = sp.rl; const LzDpResult& rr = sp.rr;
            // Is the cached DP the one the reference would run now?  Rectangles the two one-sided
            // DPs explored, +-2 cells (target rows x query columns):
            const s64 lr0 = (s64)sp.a1 + 1 - (s64)rl.max_row - 2, lr1 = (s64)sp.a1 + 2;
            const s64 lc0 = (s64)sp.a2 + 1 - (s64)rl.max_col - 2, lc1 = (s64)sp.a2 + 1 - (s64)rl.min_col + 2;
            const s64 rr0 = (s64)sp.a1 - 2, rr1 = (s64)sp.a1 + (s64)rr.max_row + 2;
            const s64 rc0 = (s64)sp.a2 + (s64)rr.min_col - 2, rc1 = (s64)sp.a2 + (s64)rr.max_col + 2;
            // (only alignments that overlap the two rectangles' rows can touch them: obi is ordered by pos1 and obi_maxend is
            // the running maximum of end1, so they are a stretch of obi found by bisection -- walking every alignment
            // committed since the snapshot was quadratic, 10 ms per strand at the north star's size)
            const Rect2 rects = { lr0, lr1, lc0, lc1, rr0, rr1, rc0, rc1 };
            auto touched_from = [&](size_t k0) { return touched_since(S, k0, rects); };
            // (a) same neighbour segments at the anchor as when it ran and nothing committed since
            //     touches what it explored: identical inputs wherever the DP looked;
            // (b) or no alignment at all touches what it explored: every bound (L, R, masks) the
            //     reference would track lies outside the band on every row, whichever neighbours it
            //     starts from, so the DP is the unconstrained one.
            bool same = nb.la == sp.nb.la && nb.ls == sp.nb.ls && nb.ra == sp.nb.ra && nb.rs == sp.nb.rs;
            const double tq1 = prof ? now() : 0;
            if (same) same = !touched_from(sp.n_snap);
            else      same = !touched_from(0);
            if (prof) t_c_chk += now() - tq1;
            const double tq2 = prof ? now() : 0;
            struct Lap { double& acc; double t0; bool on; std::function<double()> clk; ~Lap() { if (on) acc += clk() - t0; } } lap_build{ t_c_build, tq2, prof, now };
            if (!same) { cache.erase(aix); next = aix; cut = true; st.reruns++; break; }
            st.anchors_extended++;
            st.dp_cells += rl.cells + rr.cells;
            st.truncated += (rl.truncated ? 1 : 0) + (rr.truncated ? 1 : 0);     // :3640-3661: the reference warns on stderr
            Prebuilt& pb = prebuilt[spec_of[e]];
            if (!pb.have) {
                splice_and_trim(G, sp.a1, sp.a2, rl, sp.ol, rr, sp.orr, pb.b); format_segments(pb.b, pb.segs);
                if (entries[e].near_slot < slot_members.size()) for (const Member& mb : slot_members[entries[e].near_slot]) if (on_pieces(pb.segs, mb.pos1, mb.pos2)) covered[mb.e] = 1;
            }
            Built& b = pb.b;
            std::vector<LzDpSeg>& segs = pb.segs;
            cache.erase(aix);
            if (segs.empty()) continue;                        // empty alignment, :1401-1405
            if (!G.all_bounds && b.s < G.score_thresh) continue;     // :1419-1429
            LzDpAlign m; memset(&m, 0, sizeof(m));
            m.pos1 = b.start1; m.pos2 = b.start2; m.end1 = b.stop1; m.end2 = b.stop2;
            m.first_seg = (s32)S.segs.size(); m.last_seg = m.first_seg + (s32)segs.size() - 1;
            const double tq3 = prof ? now() : 0;
            align_left_right(S, m);
            if (prof) t_c_alr += now() - tq3;
            S.segs.insert(S.segs.end(), segs.begin(), segs.end());
            S.aligns.push_back(m);
            Info in; in.s = b.s; in.beg1 = b.start1 + 1; in.beg2 = b.start2 + 1; in.end1 = b.stop1 + 1; in.end2 = b.stop2 + 1;
            in.script.swap(b.script);
            info.push_back(std::move(in));
            insert_align(S, (s32)S.aligns.size() - 1);
            const double tq4 = prof ? now() : 0;
            if (prof) t_c_ins += tq4 - tq3;
            struct LapC { double& acc; double t0; bool on; std::function<double()> clk; ~LapC() { if (on) acc += clk() - t0; } } lap_cov{ t_c_cov, tq4, prof, now };
            if (entries[e].near_slot < slot_align.size()) {
                const u32 slot = entries[e].near_slot;
                slot_align[slot] = (s32)S.aligns.size() - 1;
            }
            if (G.max_paired_bases) {                          // count_paired_bases, :5695-5706; the limit test of :1441-1459
                for (const LzDpSeg& g : segs) if (g.type == LZ_DIAG_SEG) paired_bases += (u64)g.e1 + 1 - g.b1;
                if (paired_bases > G.max_paired_bases) return LZGPU_NH_PAIRED_LIMIT;
            }
        }
        if (cut) for (size_t e = 0; e < entries.size(); e++)    // for the scan of the next window
            if (!entries[e].speculated && entries[e].near_slot < slot_align.size()) near_align[entries[e].anchor_ix] = slot_align[entries[e].near_slot];
        if (!cut) next = j;
        lap(t_commit);
    }
    if (prof) fprintf(stderr, "[lzgpu hostprof] gapped: sort %.2f ms, windows %.2f ms, DP launches %.2f ms, pre-build %.2f ms, commit %.2f ms (neighbours %.2f, validity %.2f, build %.2f: of which neighbours of the new alignment + lists %.2f [align_left_right %.2f], coverage of deferred anchors %.2f) (%u rounds)\n",
                      t_sort, t_window, t_exec, t_prebuild, t_commit, t_c_lr, t_c_chk, t_c_build, t_c_ins, t_c_alr, t_c_cov, (unsigned)st.rounds);

    // ---- inhibitTrivial's test by sequence name (:1485-1545) cannot be made here: a result that holds a candidate for it
    // (one diagonal piece covering a whole partition pair of equal length, base for base the same) goes back undone
    if (delayed_check)
        for (s32 ai : S.obi) {
            const LzDpAlign& al = S.aligns[ai];
            if (info[ai].s < G.score_thresh || al.first_seg != al.last_seg || S.segs[al.first_seg].type != LZ_DIAG_SEG) continue;
            u32 lo1, hi1, lo2, hi2;
            if (!partition_limits(G.sep1, G.n_sep1, al.pos1, G.tlen, lo1, hi1) || !partition_limits(G.sep2, G.n_sep2, al.pos2, G.qlen, lo2, hi2)) continue;
            if (hi1 - lo1 != hi2 - lo2 || al.end1 + 1 - al.pos1 != hi1 - lo1) continue;
            if (memcmp(G.t + al.pos1, G.q + al.pos2, al.end1 + 1 - al.pos1) == 0) return LZGPU_NH_IDENTICAL;
        }
    // ---- output in increasing start order (orderBegInc), :1475-1566
    for (s32 ai : S.obi) {
        const Info& in = info[ai];
        if (in.s < G.score_thresh) continue;
        if (G.inhibit_trivial && in.trivial) continue;            // :1483
        lz_align a; a.beg1 = in.beg1; a.beg2 = in.beg2; a.end1 = in.end1; a.end2 = in.end2; a.s = in.s;
        a.script_len = (u32)in.script.size(); a.script_off = (u32)out_ops.size();
        out_ops.insert(out_ops.end(), in.script.begin(), in.script.end());
        out.push_back(a);
    }
    return 0;
}

int lzh_selftest_neighbours(u32 seed, u32 n_aligns, u32 n_queries)
{
    u64 x = (u64)seed * 0x9E3779B97F4A7C15ull + 3;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (u32)(x >> 12); };
    LzHostSnapshot S;
    const u32 span = 200000;
    for (u32 a = 0; a < n_aligns; a++) {                        // alignments of diagonal / vertical pieces, overlapping freely
        LzDpAlign m; memset(&m, 0, sizeof(m));
        u32 p1 = rnd() % span, p2 = rnd() % span;
        m.pos1 = p1; m.pos2 = p2; m.first_seg = (s32)S.segs.size();
        const u32 pieces = 1 + rnd() % 5;
        for (u32 k = 0; k < pieces; k++) {
            LzDpSeg g; const u32 len = 1 + rnd() % (rnd() % 8 == 0 ? 5000 : 300);
            if (k % 2 == 0) { g.type = LZ_DIAG_SEG; g.b1 = p1; g.b2 = p2; g.e1 = p1 + len - 1; g.e2 = p2 + len - 1; p1 += len; p2 += len; }
            else            { g.type = LZ_VERT_SEG; g.b1 = p1; g.b2 = p2; g.e1 = p1 + len - 1; g.e2 = p2; p1 += len; }
            S.segs.push_back(g);
        }
        m.last_seg = (s32)S.segs.size() - 1; m.end1 = p1 - 1; m.end2 = p2 - 1;
        S.aligns.push_back(m);
        insert_align(S, (s32)S.aligns.size() - 1);
    }
    int bad = 0;
    for (u32 k = 0; k < n_queries; k++) {
        u32 p1, p2;
        if (k % 3 == 0 && !S.aligns.empty()) { const LzDpSeg& g = S.segs[rnd() % S.segs.size()]; p1 = g.b1 + rnd() % (g.e1 - g.b1 + 1); p2 = g.b2 + (rnd() % 7) - 3; }
        else { p1 = rnd() % (span + 6000); p2 = rnd() % (span + 6000); }
        Neighbours a, b;
        const int ra = msp_left_right(S, p1, p2, a), rb = msp_left_right_plain(S, p1, p2, b);
        if (ra != rb || (ra == 1 && (a.la != b.la || a.ls != b.ls || a.ra != b.ra || a.rs != b.rs))) bad++;
        // the other bisected searches of the commit pass against the reference's linear walks
        s32 b1, a1, b2, a2;
        above_below(S, p1, b1, a1); above_below_plain(S, p1, b2, a2);
        if (b1 != b2 || a1 != a2) bad++;
        LzDpAlign m1; memset(&m1, 0, sizeof(m1));
        m1.pos1 = p1; m1.pos2 = p2; m1.end1 = p1 + rnd() % 4000; m1.end2 = p2 + rnd() % 4000;
        LzDpAlign m2 = m1;
        align_left_right(S, m1); align_left_right_plain(S, m2);
        if (memcmp(&m1, &m2, sizeof(m1)) != 0) bad++;
        const s64 r0 = (s64)p1 - (s64)(rnd() % 3000), r1 = (s64)p1 + 2, c0 = (s64)p2 - (s64)(rnd() % 3000), c1 = (s64)p2 + 2;
        const Rect2 q = { r0, r1, c0, c1, (s64)p1 - 2, (s64)p1 + (s64)(rnd() % 3000), (s64)p2 - 2, (s64)p2 + (s64)(rnd() % 3000) };
        const size_t k0 = S.aligns.empty() ? 0 : rnd() % S.aligns.size();
        if (touched_since(S, k0, q) != touched_since_plain(S, k0, q)) bad++;
    }
    return bad;
}

