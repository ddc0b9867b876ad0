// lz_dp_pieces.cpp -- the earlier alignments as a one-sided DP meets them, worked out per job on the HOST (round 5; LzDpPiece,
// lz_dp_dev.hpp).  Host-only code (no HIP): liblzgpu.so and tests/emul link it.
//
// While it sweeps, the reference keeps (1) the bounding segment left and right of the sweep -- it steps along an alignment's segments as
// the rows go by and hops to the alignment's recorded neighbour when it ends (update_LR_bounds, src/gapped_extend.c:4588-4700;
// next_sweep_seg / prev_sweep_seg :4754-4850) -- and (2) the list of segments crossing the sweep row, whose cells it masks
// (update_active_segs :4885-4965, build_active_seg :4992-5035).  Neither reads anything the sweep computes: both are functions of
// (anchor, direction, neighbours at the anchor, alignments committed so far).  Rounds 1-4 ran the two routines on the device, one wave
// stepping through pointers turned indices, row by row.  Here they are evaluated SEGMENT BY SEGMENT into run-length pieces of rows,
// once per job, and the kernel reads pieces (lz_dp_run): a cursor per bound, and the lanes stamp the mask cells of a row in parallel.
//
// What must be kept exactly is the reference's bookkeeping at the seams:
//   * a bound takes its new column ON the row where its old segment stops covering the sweep -- the next segment of the same alignment
//     (a horizontal one skipped), else the segment the finished alignment recorded as its neighbour at its end -- and moves on from
//     there on the following rows; when no neighbour is recorded the bound is gone for the rest of the sweep;
//   * an alignment enters the mask list on the row of its first (reversed: last) base, in list order, and only while the list's head is
//     not behind the sweep; a horizontal segment masks a run of cells on the row where the segment after it begins;
//   * columns are unsigned in the reference: left of column 0 they fall outside every band (the kernel skips negative columns).
#include <algorithm>
#include "lz_gapped_host.hpp"

#define LZ_SD(a, b) (((s32)(a)) - ((s32)(b)))
namespace {
struct Walk {
    const LzHostSnapshot& S; const LzDpJob& J; const u32 H;
    u32 a1() const { return J.anchor1; }
    u32 a2() const { return J.anchor2; }
    // the last row on which segment g covers the sweep row (forward: e1 >= row + a1; reversed: b1 <= a1 - row)
    s64 cover_end(const LzDpSeg& g) const { return J.reversed ? (s64)a1() - (s64)g.b1 : (s64)g.e1 - (s64)a1(); }

    // the bound leaves segment `seg` of alignment `al` on row `row`: where it goes, and its column there (:4754-4850)
    s32 hop(int look_right, s32& seg, s32& al, u32 row) const
    {
        const LzDpAlign& A = S.aligns[al];
        if (!J.reversed) {
            seg = (seg < A.last_seg) ? seg + 1 : -1;
            if (seg >= 0) {
                if (S.segs[seg].type == LZ_HORZ_SEG) seg = (seg < A.last_seg) ? seg + 1 : -1;
                return seg >= 0 ? LZ_SD(S.segs[seg].b2, a2()) : 0;
            }
            seg = look_right ? A.right_seg2 : A.left_seg2; al = look_right ? A.right_align2 : A.left_align2;
            if (seg < 0) return 0;
            const LzDpSeg& g = S.segs[seg];
            return g.type == LZ_DIAG_SEG ? (s32)row + LZ_SD(g.b2, a2()) - LZ_SD(g.b1, a1()) : LZ_SD(g.b2, a2());
        }
        seg = (seg > A.first_seg) ? seg - 1 : -1;
        if (seg >= 0) {
            if (S.segs[seg].type == LZ_HORZ_SEG) seg = (seg > A.first_seg) ? seg - 1 : -1;
            return seg >= 0 ? LZ_SD(a2(), S.segs[seg].e2) : 0;
        }
        seg = look_right ? A.right_seg1 : A.left_seg1; al = look_right ? A.right_align1 : A.left_align1;
        if (seg < 0) return 0;
        const LzDpSeg& g = S.segs[seg];
        return g.type == LZ_DIAG_SEG ? (s32)row + LZ_SD(a2(), g.e2) - LZ_SD(a1(), g.e1) : LZ_SD(a2(), g.e2);
    }

    // One bound as pieces of rows 1, 2, ...  (seg, al): the segment it starts on; v: its column on row 0 (the sweep's set-up); adj: +1 for
    // the bound that becomes L, -1 for the one that becomes R.  -> false if the horizon cut the chain (more rows would need more pieces)
    bool bound(s32 seg, s32 al, s32 v, int look_right, s32 adj, std::vector<LzDpPiece>& out) const
    {
        u32 row = 0;
        while (seg >= 0) {
            if (row >= H) return false;
            const LzDpSeg& g = S.segs[seg];
            // (backwards the reference tests b1 <= a1 - row in unsigned arithmetic: on the one row a sweep can have beyond the target's
            // first base, row a1 + 1, the difference wraps and whatever segment is in force "covers" it)
            const s64 ce = (J.reversed && row >= a1()) ? (s64)H : cover_end(g);
            if (ce > (s64)row) {                                   // rows row + 1 .. ce: this segment, one column further per row if diagonal
                const u32 slope = g.type == LZ_DIAG_SEG ? 1u : 0u;
                const u32 r1 = (u32)std::min<s64>(ce, (s64)H);
                out.push_back(LzDpPiece{ row + 1, r1, v + (s32)slope, slope });
                v += (s32)(slope * (r1 - row)); row = r1;
                continue;
            }
            row++;                                                 // this row the segment no longer covers: the bound hops and takes its new column
            v = hop(look_right, seg, al, row) + adj;
            if (seg >= 0) out.push_back(LzDpPiece{ row, row, v, 0u });
        }
        return true;
    }

    // One entry of the reference's list of active segments, built on row cr (build_active_seg, :4992-5035): a diagonal or vertical segment
    // masks one cell per row from cr to its last row; a horizontal one a run of cells on row cr (and, were it to stay in the list, the
    // run's first cell on the rows that follow).  -> the segment's last row
    s64 entry(s32 seg, u32 cr, std::vector<LzDpPiece>& out) const
    {
        const LzDpSeg& g = S.segs[seg];
        const s32 x = J.reversed ? LZ_SD(a2(), g.e2) : LZ_SD(g.b2, a2());
        const s64 last = std::min<s64>(J.reversed ? (s64)a1() - (s64)g.b1 : (s64)g.e1 - (s64)a1(), 0xFFFFFFF0ll);
        if (g.type != LZ_HORZ_SEG) { out.push_back(LzDpPiece{ cr, (u32)std::max<s64>(last, cr), x, g.type == LZ_DIAG_SEG ? 1u : 0u }); return last; }
        const s32 xe = J.reversed ? LZ_SD(a2(), g.b2) : LZ_SD(g.e2, a2());
        if (xe >= x) out.push_back(LzDpPiece{ cr, cr, x, (u32)(xe - x) << 1 });
        if (last > (s64)cr) out.push_back(LzDpPiece{ cr + 1, (u32)last, x, 0u });
        return last;
    }
    // the segments of alignment ai from the row it enters the list on (update_active_segs, :4885-4965); -> false if the horizon cut it short
    bool seg_has_next(const LzDpAlign& A, s32 seg) const { return J.reversed ? seg > A.first_seg : seg < A.last_seg; }
    bool masks_of(s32 ai, u32 row, std::vector<LzDpPiece>& out) const
    {
        const LzDpAlign& A = S.aligns[ai];
        s32 seg = J.reversed ? A.last_seg : A.first_seg;
        u32 cr = row;
        s64 last = entry(seg, cr, out);
        for (;;) {
            const s64 e = std::max<s64>(last, cr) + 1;             // the first row the entry's segment no longer reaches: the list moves on
            if (e > (s64)H) return seg_has_next(A, seg) ? false : true;   // (rows beyond the horizon: more pieces only if segments remain)
            seg = J.reversed ? ((seg > A.first_seg) ? seg - 1 : -1) : ((seg < A.last_seg) ? seg + 1 : -1);
            if (seg < 0) return true;                              // the alignment is behind the sweep
            cr = (u32)e;
            if (S.segs[seg].type == LZ_HORZ_SEG) {                 // its run of cells on this row, and straight on to the segment behind it
                const LzDpSeg& g = S.segs[seg];
                const s32 x = J.reversed ? LZ_SD(a2(), g.e2) : LZ_SD(g.b2, a2()), xe = J.reversed ? LZ_SD(a2(), g.b2) : LZ_SD(g.e2, a2());
                if (xe >= x) out.push_back(LzDpPiece{ cr, cr, x, (u32)(xe - x) << 1 });
                seg = J.reversed ? seg - 1 : seg + 1;              // (a horizontal piece is never terminal)
                if (seg < A.first_seg || seg > A.last_seg) return true;
            }
            last = entry(seg, cr, out);
        }
    }
};
}

void lzh_dp_pieces(const LzHostSnapshot& S, const LzDpJob& J, u32 horizon, LzDpPieces& out)
{
    out.lb.clear(); out.rb.clear(); out.mk.clear(); out.complete = true;
    if (S.aligns.empty()) return;
    const Walk w{ S, J, horizon };
    // ---- the bounds on row 0 (ydrop_one_sided_align's set-up, :3520-3541: a diagonal segment is extended down to the anchor's row)
    s32 L = 0, R = (s32)J.N + 1;
    if (J.left_seg >= 0)  { const LzDpSeg& g = S.segs[J.left_seg];  L = LZ_SD(g.b2, J.anchor2); if (g.type == LZ_DIAG_SEG) L -= LZ_SD(g.b1, J.anchor1); }
    if (J.right_seg >= 0) { const LzDpSeg& g = S.segs[J.right_seg]; R = LZ_SD(g.b2, J.anchor2); if (g.type == LZ_DIAG_SEG) R -= LZ_SD(g.b1, J.anchor1); }
    if (!J.reversed) {
        if (J.left_seg >= 0  && !w.bound(J.left_seg,  J.left_align,  L, 0, +1, out.lb)) out.complete = false;
        if (J.right_seg >= 0 && !w.bound(J.right_seg, J.right_align, R, 1, -1, out.rb)) out.complete = false;
    } else {
        // the backward sweep mirrors the columns: what was right of the anchor bounds it on the left (note 14, :3536-3541)
        if (J.left_seg < 0 && J.right_seg >= 0)       { L = -R + 1; R = (s32)J.N + 1; }
        else if (J.left_seg >= 0 && J.right_seg < 0)  { R = -L - 1; L = 0; }
        else if (J.left_seg >= 0 && J.right_seg >= 0) { const s32 t = -L - 1; L = -R + 1; R = t; }
        if (J.right_seg >= 0 && !w.bound(J.right_seg, J.right_align, L, 1, +1, out.lb)) out.complete = false;
        if (J.left_seg >= 0  && !w.bound(J.left_seg,  J.left_align,  R, 0, -1, out.rb)) out.complete = false;
    }
    // ---- the alignments the sweep will reach, in the list's order (aboveList by increasing start / belowList by decreasing end)
    const std::vector<s32>& order = J.reversed ? S.oed : S.obi;
    u32 fired = 1;                                               // an alignment enters on the row that EQUALS its first row: rows come 1, 2, ...
    for (s64 pos = J.list_start; pos >= 0 && pos < (s64)order.size(); pos++) {
        const LzDpAlign& A = S.aligns[order[(size_t)pos]];
        const u32 r = J.reversed ? J.anchor1 - A.end1 : A.pos1 - J.anchor1;      // (unsigned, as the reference has it)
        if (r < fired) break;                                    // the head of the list is behind the sweep: nothing enters any more
        if (r > horizon) { out.complete = false; break; }
        if (!w.masks_of(order[(size_t)pos], r, out.mk)) out.complete = false;
        fired = r;
    }
    std::stable_sort(out.mk.begin(), out.mk.end(), [](const LzDpPiece& a, const LzDpPiece& b) { return a.r0 < b.r0; });
}
