// emul_bounds_plain.cpp -- TEST INFRASTRUCTURE: the yardstick for lz_dp_pieces.cpp.  The reference keeps a DP's bounds and its list of
// active segments up to date ROW BY ROW while it sweeps (update_LR_bounds, src/gapped_extend.c:4588-4700; next_sweep_seg / prev_sweep_seg
// :4754-4850; update_active_segs :4885-4965; build_active_seg :4992-5035).  This file restates those routines as they stand -- pointers
// as indices, one call per row -- and emul_verify_pieces() compares, row by row, what they yield with what the product's pieces yield
// (the product evaluates the same bookkeeping segment by segment, on the host, once per job).  Never part of liblzgpu.so.
#include <stdio.h>
#include <vector>
#include <algorithm>
#include "../../lastz_amd/csrc/lz_gapped_host.hpp"

#define SDIFF(a, b) (((s32)(a)) - ((s32)(b)))
namespace {
struct Act { s32 align, seg; u32 x, last_row; s32 type; s32 filter; };
struct Plain {
    const LzHostSnapshot& S; const LzDpJob& J;
    s32 L, R; s32 left_seg, right_seg, left_align, right_align, list_pos; u32 next_act_row;
    std::vector<Act> act;
    std::vector<std::pair<u32, u32>> stamped;                  // the cells masked on the current row, as [from, to] runs of unsigned columns

    s32 next_sweep_seg(int look_right, s32& seg, s32& al, u32 row) const
    {
        const u32 a1 = J.anchor1, a2 = J.anchor2;
        seg = (seg < S.aligns[al].last_seg) ? seg + 1 : -1;
        if (seg >= 0) {
            if (S.segs[seg].type == LZ_HORZ_SEG) seg = (seg < S.aligns[al].last_seg) ? seg + 1 : -1;
            if (seg >= 0) return SDIFF(S.segs[seg].b2, a2);
            return 0;
        }
        if (look_right) { seg = S.aligns[al].right_seg2; al = S.aligns[al].right_align2; }
        else            { seg = S.aligns[al].left_seg2;  al = S.aligns[al].left_align2; }
        if (seg < 0) return 0;
        if (S.segs[seg].type == LZ_DIAG_SEG) return (s32)row + SDIFF(S.segs[seg].b2, a2) - SDIFF(S.segs[seg].b1, a1);
        return SDIFF(S.segs[seg].b2, a2);
    }
    s32 prev_sweep_seg(int look_right, s32& seg, s32& al, u32 row) const
    {
        const u32 a1 = J.anchor1, a2 = J.anchor2;
        seg = (seg > S.aligns[al].first_seg) ? seg - 1 : -1;
        if (seg >= 0) {
            if (S.segs[seg].type == LZ_HORZ_SEG) seg = (seg > S.aligns[al].first_seg) ? seg - 1 : -1;
            if (seg >= 0) return SDIFF(a2, S.segs[seg].e2);
            return 0;
        }
        if (look_right) { seg = S.aligns[al].right_seg1; al = S.aligns[al].right_align1; }
        else            { seg = S.aligns[al].left_seg1;  al = S.aligns[al].left_align1; }
        if (seg < 0) return 0;
        if (S.segs[seg].type == LZ_DIAG_SEG) return (s32)row + SDIFF(a2, S.segs[seg].e2) - SDIFF(a1, S.segs[seg].e1);
        return SDIFF(a2, S.segs[seg].e2);
    }
    void setup()
    {
        L = 0; R = (s32)J.N + 1;
        if (J.left_seg >= 0)  { const LzDpSeg& g = S.segs[J.left_seg];  L = SDIFF(g.b2, J.anchor2); if (g.type == LZ_DIAG_SEG) L -= SDIFF(g.b1, J.anchor1); }
        if (J.right_seg >= 0) { const LzDpSeg& g = S.segs[J.right_seg]; R = SDIFF(g.b2, J.anchor2); if (g.type == LZ_DIAG_SEG) R -= SDIFF(g.b1, J.anchor1); }
        if (J.reversed) {
            if (J.left_seg < 0 && J.right_seg >= 0)       { L = -R + 1; R = (s32)J.N + 1; }
            else if (J.left_seg >= 0 && J.right_seg < 0)  { R = -L - 1; L = 0; }
            else if (J.left_seg >= 0 && J.right_seg >= 0) { s32 t = -L - 1; L = -R + 1; R = t; }
        }
        left_align = J.left_align; right_align = J.right_align; left_seg = J.left_seg; right_seg = J.right_seg;
        list_pos = J.list_start;
        peek();
    }
    void peek()
    {
        if (list_pos < 0 || list_pos >= (s32)S.aligns.size()) { list_pos = -1; next_act_row = 0xFFFFFFFFu; return; }
        const std::vector<s32>& order = J.reversed ? S.oed : S.obi;
        const LzDpAlign& al = S.aligns[order[list_pos]];
        next_act_row = J.reversed ? (J.anchor1 - al.end1) : (al.pos1 - J.anchor1);
    }
    void update_lr(u32 row)
    {
        const u32 a1 = J.anchor1;
        if (!J.reversed) {
            if (left_seg >= 0) {
                if (S.segs[left_seg].e1 >= row + a1) { if (S.segs[left_seg].type == LZ_DIAG_SEG) L++; }
                else L = next_sweep_seg(0, left_seg, left_align, row) + 1;
            }
            if (right_seg >= 0) {
                if (S.segs[right_seg].e1 >= row + a1) { if (S.segs[right_seg].type == LZ_DIAG_SEG) R++; }
                else R = next_sweep_seg(1, right_seg, right_align, row) - 1;
            }
        } else {
            if (right_seg >= 0) {
                if (S.segs[right_seg].b1 <= a1 - row) { if (S.segs[right_seg].type == LZ_DIAG_SEG) L++; }
                else L = prev_sweep_seg(1, right_seg, right_align, row) + 1;
            }
            if (left_seg >= 0) {
                if (S.segs[left_seg].b1 <= a1 - row) { if (S.segs[left_seg].type == LZ_DIAG_SEG) R++; }
                else R = prev_sweep_seg(0, left_seg, left_align, row) - 1;
            }
        }
    }
    void stamp(u32 x) { stamped.push_back({ x, x }); }
    void build_active(Act& a, u32 row)
    {
        const LzDpSeg& sg = S.segs[a.seg];
        a.type = sg.type;
        if (!J.reversed) { a.x = sg.b2 - J.anchor2; a.last_row = sg.e1 - J.anchor1; }
        else             { a.x = J.anchor2 - sg.e2; a.last_row = J.anchor1 - sg.b1; }
        if (a.type != LZ_HORZ_SEG) stamp(a.x);
        else {
            const u32 horz_end = (!J.reversed) ? sg.e2 - J.anchor2 : J.anchor2 - sg.b2;
            if (a.x <= horz_end) stamped.push_back({ a.x, horz_end });          // (the band's clip is the caller's)
        }
        (void)row;
    }
    void update_active(u32 row)
    {
        stamped.clear();
        for (size_t k = 0; k < act.size(); k++) {
            Act& a = act[k];
            if (a.last_row >= row) { if (a.type == LZ_DIAG_SEG) a.x++; stamp(a.x); }
            else {
                const LzDpAlign& al = S.aligns[a.align];
                const s32 nx = J.reversed ? ((a.seg > al.first_seg) ? a.seg - 1 : -1) : ((a.seg < al.last_seg) ? a.seg + 1 : -1);
                if (nx >= 0) {
                    a.seg = nx; build_active(a, row);
                    if (a.type == LZ_HORZ_SEG) { a.seg = J.reversed ? a.seg - 1 : a.seg + 1; build_active(a, row); }
                } else a.filter = 1;
            }
        }
        while (list_pos >= 0 && next_act_row == row) {
            const std::vector<s32>& order = J.reversed ? S.oed : S.obi;
            const LzDpAlign& al = S.aligns[order[list_pos]];
            Act a; a.filter = 0; a.align = order[list_pos]; a.seg = J.reversed ? al.last_seg : al.first_seg;
            build_active(a, row);
            act.push_back(a);
            list_pos++; peek();
        }
        act.erase(std::remove_if(act.begin(), act.end(), [](const Act& a) { return a.filter != 0; }), act.end());
    }
};
}

// rows 1 .. rows of job J against snapshot S: the product's pieces (asked for up to `horizon`) against the row-by-row routines.
// -> 0, or the first row that differs (and a line on stderr)
u32 emul_verify_pieces(const LzHostSnapshot& S, const LzDpJob& J, u32 horizon, u32 rows)
{
    LzDpPieces P; lzh_dp_pieces(S, J, horizon, P);
    if (!P.complete && rows > horizon) rows = horizon;
    Plain pl{ S, J }; pl.setup();
    size_t lbi = 0, rbi = 0;
    s32 L = 0, R = (s32)J.N + 1;
    for (u32 row = 1; row <= rows; row++) {
        pl.update_lr(row); pl.update_active(row);
        // the product's row: the kernel's cursor logic (lz_dp_run)
        while (lbi < P.lb.size() && row > P.lb[lbi].r1) { lbi++; if (lbi == P.lb.size()) L = 1; }
        if (lbi < P.lb.size()) { if (row < P.lb[lbi].r0) { fprintf(stderr, "verify: left pieces not contiguous at row %u\n", row); return row; } L = lz_dp_piece_at(P.lb[lbi], row); }
        while (rbi < P.rb.size() && row > P.rb[rbi].r1) { rbi++; if (rbi == P.rb.size()) R = -1; }
        if (rbi < P.rb.size()) { if (row < P.rb[rbi].r0) { fprintf(stderr, "verify: right pieces not contiguous at row %u\n", row); return row; } R = lz_dp_piece_at(P.rb[rbi], row); }
        const bool has_l = lbi < P.lb.size(), has_r = rbi < P.rb.size();
        const bool pl_l = J.reversed ? pl.right_seg >= 0 : pl.left_seg >= 0, pl_r = J.reversed ? pl.left_seg >= 0 : pl.right_seg >= 0;
        if (has_l != pl_l || has_r != pl_r || (has_l && L != pl.L) || (pl_r && R != pl.R) || (!has_r && P.rb.size() && R != pl.R)) {
            fprintf(stderr, "verify: bounds differ at row %u (rev %d): pieces L %d%s R %d%s, row by row L %d%s R %d%s\n", row, J.reversed, L, has_l ? "" : "(gone)", R, has_r ? "" : "(gone)",
                    pl.L, pl_l ? "" : "(gone)", pl.R, pl_r ? "" : "(gone)");
            return row;
        }
        std::vector<std::pair<u32, u32>> mine;
        for (const LzDpPiece& pc : P.mk) {
            if (row < pc.r0 || row > pc.r1) continue;
            const s32 xs = lz_dp_piece_at(pc, row);
            if (xs < 0) continue;
            mine.push_back({ (u32)xs, (u32)xs + (pc.fl >> 1) });
        }
        std::vector<std::pair<u32, u32>> theirs;
        for (auto& r : pl.stamped) if (r.first < 0x80000000u) theirs.push_back(r);      // (an unsigned column left of 0 is outside every band)
        std::sort(mine.begin(), mine.end()); std::sort(theirs.begin(), theirs.end());
        if (mine != theirs) {
            fprintf(stderr, "verify: masked cells differ at row %u (rev %d, horizon %u): pieces %zu runs, row by row %zu runs\n", row, J.reversed, horizon, mine.size(), theirs.size());
            return row;
        }
    }
    return 0;
}
