// lz_dp_dev.hpp -- the Y-drop gapped extension (ydrop_one_sided_align, src/gapped_extend.c:3388-3868)
// as ONE WORKGROUP OF LZ_DP_LANES LANES PER ONE-SIDED DP, written once for the device and for the host
// test harness.
//
// Execution model.  lz_dp_run() is a sequence of steps over the lanes 0..LZ_DP_LANES-1, which only
// communicate through the LzDpShared block (LDS on the GPU) and through the executor's cross-lane
// steps.  The executor X supplies:
//   X::phase(f)   f(lane) on every lane, then a barrier;
//   X::step(f)    the same without a barrier (f touches registers and the lane's own LDS cells only);
//   X::leader(f)  a serial piece, once per DP, then a barrier (on the GPU: one whole wave in lockstep,
//                 its inputs passed through X::uni so that the sweep state LzDpCtl stays scalar);
//   X::scan_gap / scan_cand / reduce_row / row_result   the cross-lane steps (DPP scans + LDS partials
//                 on the GPU, plain loops in tests/emul).
// Control flow between steps only depends on values every lane reads from LzDpShared after a barrier,
// so it is uniform by construction.
//
// Row algorithm.  The reference sweeps a row left to right with three
// loop-carried values: the insertion score i, the running bestScore (for the Y-drop test) and the
// left bound LY.  All three have closed parallel forms:
//   * i is a max-plus prefix scan: i[c+1] = max(open[c], i[c]-gapE), open[c] = C_diag[c]-gapOE
//     unless the cell is masked or its D beats the diagonal (no gap may open after a gap,
//     :3714-3727).  Cells the reference prunes reset i to -inf; leaving such values in the scan is
//     harmless because every value that differs is below the (non-decreasing) prune threshold.
//   * the running bestScore before column c is max(best_at_row_start, max of diagonal-won cells
//     left of c): a plain prefix max, because a cell that raises the best is never pruned.
//   * LY advances over the leading pruned cells: LY' = first live column.
// So a row is: leader step (row end of the previous row, :3786-3827, + bounds / masks / traceback budget
// of this one) -> walk 1 (per-lane block summaries of the i-recurrence) -> scan -> walk 2 (cells, links,
// candidate bests) -> prefix max -> walk 3 (prune test, stores, traceback bytes) -> row reduction.
// Lane l owns cpl = ceil(width/LZ_DP_LANES) consecutive columns of the row.
#pragma once
#include <type_traits>
#include "lz_common.hpp"

#ifndef LZ_DP_LANES
#define LZ_DP_LANES   256             // lanes (threads) per one-sided DP: 4 waves of one workgroup
#endif
#define LZ_DP_WAVES   (LZ_DP_LANES / 64)
#ifndef LZ_DP_MAXW
#define LZ_DP_MAXW    2048            // ring size (columns) of the sweep row held in LDS
#endif
#define LZ_DP_WIDEW   65536           // ... of the wide variant, whose ring lives in HBM (bands the LDS ring cannot hold)
#define LZ_DP_TBWIN   64              // traceback look-ahead window (links along one diagonal)
#ifndef LZ_DP_BATCH
#define LZ_DP_BATCH   2               // cells whose LDS reads are issued together in the walks (and a lane's first cells, carried in registers from walk to walk)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define LZ_UNROLL _Pragma("unroll")
#else
#define LZ_UNROLL
#endif
#define LZ_DP_SERIAL_FILL 4           // overhang cells lane 0 stores itself at row end
#ifndef LZ_DP_STAMP_PERIOD
#define LZ_DP_STAMP_PERIOD 65535        // rows after which the 16-bit mask stamps of the LDS ring start over (tests: a small period)
#endif
#define LZ_DP_NEGINF  ((s32)-1932735283)      // negInfinity, src/dna_utilities.h:138

enum { LZ_DIAG_SEG = 0, LZ_HORZ_SEG = 1, LZ_VERT_SEG = 2 };
enum { LZ_C_FROM_C = 0, LZ_C_FROM_I = 1, LZ_C_FROM_D = 2, LZ_I_EXT = 4, LZ_D_EXT = 8 };
enum { LZ_DP_OK = 0, LZ_DP_TOO_WIDE = 1, LZ_DP_TB_SLOT = 2, LZ_DP_ROW_SLOT = 3, LZ_DP_OPS_SLOT = 4, LZ_DP_PIECE_SLOT = 5 };   // PIECE_SLOT: the sweep passed the job's piece horizon

struct LzDpSeg   { u32 b1, b2, e1, e2; s32 type; };
struct LzDpAlign {                      // galign, src/gapped_extend.c:222-250 (indices instead of pointers, -1 = NULL); host side (LzHostSnapshot)
    u32 pos1, pos2, end1, end2;
    s32 first_seg, last_seg;            // this alignment's segments are segs[first_seg..last_seg]
    s32 left_align1, right_align1, left_align2, right_align2;
    s32 left_seg1, right_seg1, left_seg2, right_seg2;
};

// How earlier alignments reach a one-sided DP (round 5).  The reference follows them while it sweeps: the bounding segments left and
// right of the anchor, row by row (update_LR_bounds, src/gapped_extend.c:4588), and a list of the segments that cross the sweep row,
// whose cells it masks (update_active_segs, :4885).  Both are functions of the job and of the alignments committed before it -- not of
// anything the sweep computes -- so the host works them out once per job (lz_gapped_host.cpp: lzh_dp_pieces) as PIECES of rows:
//   bound pieces   rows r0..r1: the bound is x0 + (row - r0) * (fl & 1); consecutive pieces are contiguous in rows; when the list ends
//                  the bound is gone (the reference's left_seg / right_seg == NULL)
//   mask pieces    rows r0..r1: the cells x .. x + (fl >> 1) with x = x0 + (row - r0) * (fl & 1) are masked (a diagonal or vertical
//                  segment: one cell per row; a horizontal one: a run of cells on one row); sorted by r0
// and the kernel reads the next piece when a row passes the end of the current one: no pointer chasing through alignments and segments
// on the device, no serial list update per row -- the row's mask cells are stamped by the lanes, one piece each.  Pieces are complete
// for rows <= LzDpJob::horizon; a sweep that gets further stops with LZ_DP_PIECE_SLOT and is run again with a longer horizon.
struct LzDpPiece { u32 r0, r1; s32 x0; u32 fl; };

struct LzDpJob {
    u32 anchor1, anchor2; s32 reversed; u32 M, N;
    s32 left_align, right_align, left_seg, right_seg;   // io->leftAlign.. (msp_left_right): host side, input of lzh_dp_pieces
    s32 list_start;                     // index into obi (forward: aboveList) / oed (reversed: belowList), -1 = none: host side
    u64 tb_off; u32 tb_cap;             // traceback bytes slot
    u64 row_off; u32 row_cap;           // tbRow[] slot (u32 per row)
    u64 ops_off; u32 ops_cap;           // edit ops slot (u32 each, traceback order)
    u64 pc_off;                         // the job's pieces in the launch's arena: n_lb left-bound pieces, n_rb right-bound pieces, n_mk mask pieces
    u32 n_lb, n_rb, n_mk, horizon;
    u32 est_rows;                       // host-side guess of how many rows this DP will sweep (launch order only: longest first)
    u32 problem;                        // which LzDpProblem of the launch this DP belongs to (0 in a single-problem launch)
};

// One launch can hold the DPs of several independent problems (the two strands of a query, the tweener windows of a
// strand: lzgpu_gapped_extend_batch): each has its own query / window of the sequences.
struct LzDpProblem { const u8* tdp; const u8* qdp; u32 tlen, qlen; };

struct LzDpResult {
    s32 score; u32 end1, end2; u32 n_ops; u32 status; u32 truncated;
    u32 max_row, min_col, max_col;      // explored region in DP coordinates (row 0..max_row)
    u32 tb_used; u64 cells;
    u64 t_rows, t_trace;                // shader-clock ticks spent in the row sweep / the traceback (0 off-device)
    u64 t_begin, t_end;                 // constant 100 MHz clock (s_memrealtime, the same on every CU) when the DP started / ended (LZGPU_DPPROF)
    u64 t_ph[4];                        // ... of which: lane-0 step, walk 1 + gap scan, walk 2 + best scan, walk 3 + reduce
    u64 t_ld[5];                        // the lane-0 step, split (leading wave only): row results, row end, bounds, active segments, budget + publish
};

struct LzDpParams {                     // per batch
    const u8* tdp; u32 tlen;            // DP-class codes of the target / query (unmasked scoring classes)
    const u8* qdp; u32 qlen;
    s32 gap_e, gap_oe, ydrop, ydrop_tail;
    s32 no_trim;                        // !trimToPeak: an end on the last row / column may be reported instead of the peak (:3747-3750, :3866)
    u32 tb_len;                         // the REFERENCE's traceback size (truncation rule, :3640-3661)
    u8* tb_arena; u32* row_arena; u32* ops_arena;
    const LzDpPiece* pc_arena;
};

struct LzDpGap { s32 A, K; u32 cut; };       // f(x) = cut ? A : max(A, x - K)

#if !defined(__HIP_DEVICE_COMPILE__)
void lz_dp_row16_overflow();              // (test harness only: a 16-bit cell out of range)
#endif
#define LZ_DP_ROW16_MARGIN 1024
// may a DP with these parameters keep its sweep row in 16 bits?  (tab: the n score classes x 32 the kernel will index)
inline bool lz_dp_row16_ok(s32 ydrop, s32 gap_oe, const s32* tab, u32 n)
{
    s32 mx = 0;
    for (u32 k = 0; k < n; k++) if (tab[k] > mx) mx = tab[k];
    return ydrop >= 0 && gap_oe >= 0 && (s64)ydrop + gap_oe + 1 + LZ_DP_ROW16_MARGIN + mx <= 65535;
}

// The sweep row is a ring indexed by column & (RING-1).  Two homes for it: arrays in the DP's LDS block (the
// normal kernel), or a slot in HBM behind pointers (k_ydrop_wide: the rare bands wider than the LDS ring --
// small gap-extension penalties, huge y-drops; the same code, flat loads instead of ds loads).
// Mask stamps.  The reference stamps mask[col] = row (:3706) and a cell is masked when its stamp is the current row.
// In LDS a stamp is 16 bits (the ring's LDS bytes decide how many DPs share a CU: 34 KiB -> 27 KiB per DP, four -> six
// DPs per CU): stamp(row) = 1 + (row - 1) mod 65535, never 0 (= no stamp), and the ring's stamps are cleared whenever
// the stamp wraps to 1 again (lz_dp_run, before the row's own stamps are written), so that a stamp left behind
// 65535 rows ago cannot be taken for the current row.
// The C / D cells of the ring are reached through four accessors (ld_cd, ld_c, st_cd, st_dead) that take the ring index and the BASE
// of the row the cell belongs to; the 32-bit rings ignore the base, the 16-bit ones (below) store offsets from it.
template <u32 W> struct LzDpCells32 {
    static constexpr bool ROW16 = false;
    s32 cc[W], dd[W];                     // C[row][col], D[row+1][col]
    LZ_HD void ld_cd(u32 rx, s32, s32& c, s32& d) const { c = cc[rx]; d = dd[rx]; }
    LZ_HD s32  ld_c(u32 rx, s32) const { return cc[rx]; }
    LZ_HD void st_cd(u32 rx, s32, s32 c, s32 d) { cc[rx] = c; dd[rx] = d; }
    LZ_HD void st_dead(u32 rx) { cc[rx] = LZ_DP_NEGINF; dd[rx] = LZ_DP_NEGINF; }
    LZ_HD s32& scratch(u32 k) { return cc[k]; }           // (the dead sweep row as 32-bit scratch words, k < W)
};
// The 16-bit sweep row (round 5).  Everything a row reads or writes lies in a window below the running best: with
// F(row) = best(at the start of the row) - yDrop - gapOE - 1,
//   * a C above F is stored exactly, and so is a D above F;
//   * a value v <= F can never matter again: a cell with C = v is pruned (v < best - yDrop, :3760), a gap opened from it or extended
//     from a D / I = v stays <= F, F never falls (best never does), and every comparison that decides a link of a LIVE cell has
//     C - gapOE >= best - yDrop - gapOE = F + 1 on its other side -- so any two values <= F are interchangeable, negInfinity included;
//   * no cell of the row exceeds best + (largest substitution score).
// A cell is therefore kept as a 16-bit offset from base(row) = F(row) - 1024, clamped at 0 from below (0 = "some value <= F":
// negInfinity, masked and pruned cells), C in the low and D in the high half of one word: one LDS read and one write per cell instead of
// two, and 12 instead of 20 bytes of LDS per column -- eleven DPs per CU instead of seven.  The launcher takes this kernel only when
// yDrop + gapOE + 1025 + max(score) <= 65535 (lz_dp_row16_ok); otherwise the 32-bit row.
template <u32 W> struct LzDpCells16 {
    static constexpr bool ROW16 = true;
    u32 cd[W];
    LZ_HD void ld_cd(u32 rx, s32 base, s32& c, s32& d) const { const u32 v = cd[rx]; c = (s32)(v & 0xFFFFu) + base; d = (s32)(v >> 16) + base; }
    LZ_HD s32  ld_c(u32 rx, s32 base) const { return (s32)(cd[rx] & 0xFFFFu) + base; }
    LZ_HD void st_cd(u32 rx, s32 base, s32 c, s32 d)
    {
        // (c - base cannot wrap: every cell is >= negInfinity - 2^24 and base >= -65535; written as max(x, 0) the clamp is one instruction, as a
        // compare + select it was two -- four vector instructions less per cell pair of walk 2)
        const s32 ca = c - base, da = d - base;
        const s32 a = ca > 0 ? ca : 0, b = da > 0 ? da : 0;
#if !defined(__HIP_DEVICE_COMPILE__)
        if (a > 65535 || b > 65535) lz_dp_row16_overflow();      // (test harness: the launcher's rule must make this unreachable)
#endif
        cd[rx] = (u32)a | ((u32)b << 16);
    }
    LZ_HD void st_dead(u32 rx) { cd[rx] = 0u; }
    LZ_HD s32& scratch(u32 k) { return reinterpret_cast<s32*>(cd)[k]; }
};
template <u32 W, template <u32> class Cells = LzDpCells32> struct LzDpRingLds : Cells<W> {
    static constexpr u32 RING = W;
    static constexpr bool STAMP_WRAPS = true;
    typedef unsigned short stamp_t;
    static LZ_HD u32 stamp(u32 row) { return 1u + (row - 1u) % (u32)LZ_DP_STAMP_PERIOD; }
    stamp_t mk[W];                        // mask stamps
    u8  lk[W];                            // traceback link of the current row
    u8  bb[W];                            // B (query) score classes of the band's columns
};
// the ring of a problem without earlier alignments (lz_dp_run<.., BOUNDS = false>): nothing is ever masked, no stamps --
// 4 KiB less per DP, seven DPs per CU instead of six
template <u32 W, template <u32> class Cells = LzDpCells32> struct LzDpRingLdsNoMask : Cells<W> {
    static constexpr u32 RING = W;
    static constexpr bool STAMP_WRAPS = false;
    typedef unsigned short stamp_t;
    static LZ_HD u32 stamp(u32 row) { return row; }
    stamp_t mk[1];                        // (never touched)
    u8  lk[W];
    u8  bb[W];
};
struct LzDpRingHbm {
    static constexpr u32 RING = LZ_DP_WIDEW;
    static constexpr bool STAMP_WRAPS = false;
    typedef u32 stamp_t;
    static LZ_HD u32 stamp(u32 row) { return row; }
    static constexpr bool ROW16 = false;
    s32 *cc, *dd; u32* mk; u8 *lk, *bb;
    LZ_HD void ld_cd(u32 rx, s32, s32& c, s32& d) const { c = cc[rx]; d = dd[rx]; }
    LZ_HD s32  ld_c(u32 rx, s32) const { return cc[rx]; }
    LZ_HD void st_cd(u32 rx, s32, s32 c, s32 d) { cc[rx] = c; dd[rx] = d; }
    LZ_HD void st_dead(u32 rx) { cc[rx] = LZ_DP_NEGINF; dd[rx] = LZ_DP_NEGINF; }
    LZ_HD s32& scratch(u32 k) { return cc[k]; }
    static constexpr size_t SLOT_BYTES = (size_t)LZ_DP_WIDEW * 14;
    LZ_HD void bind(u8* slot) { cc = (s32*)slot; dd = cc + LZ_DP_WIDEW; mk = (u32*)(dd + LZ_DP_WIDEW); lk = (u8*)(mk + LZ_DP_WIDEW); bb = lk + LZ_DP_WIDEW; }
};
struct LzDpSharedBase {
    // what lane 0 publishes for the other lanes before each row (its own sweep state is LzDpCtl): sixteen words,
    // written together at the end of the lane-0 step and read together behind its barrier (four 128-bit LDS
    // accesses each way instead of sixteen / ten scattered ones)
    // (the first eight are what every row needs: two 128-bit reads, issued together, BEFORE any branch on them -- a read
    // behind a branch behind a read is an LDS round trip each, ~100 cycles for a wave that has its SIMD to itself)
    alignas(16) u32 row; u32 LY, ry_iter, cpl;
    s32 best; u32 trow_cur, done, extra;
    u32 mk_lo, fill_n, fill_base, fill_trow;              // work for all lanes before the next row (mk_lo, mk_hi: the mask pieces in reach of the row)
    u32 stage_lo, stage_a; s32 fill_i; u32 b_hi;
    u32 mk_hi, pad_[3];
    u8  aa[LZ_DP_LANES];                  // A (target) score classes of a block of 64 rows
    // per-wave partials of the cross-lane steps (GPU executor) and the row results (written by lane 0)
    LzDpGap wg[LZ_DP_WAVES]; s32 wc[LZ_DP_WAVES], wcmax[LZ_DP_WAVES]; u32 wfirst[LZ_DP_WAVES], wlast[LZ_DP_WAVES], wccol[LZ_DP_WAVES], whas[LZ_DP_WAVES];
    u32 r_first, r_last, r_ccol; s32 r_cmax;
    // traceback state
    u32 tb_row, tb_col, tb_prev, tb_nops, tb_run_op, tb_run_len, tb_done;
};
template <class Ring> struct LzDpSharedT : LzDpSharedBase, Ring {};
typedef LzDpSharedT<LzDpRingLds<LZ_DP_MAXW>> LzDpShared;
typedef LzDpSharedT<LzDpRingLdsNoMask<LZ_DP_MAXW>> LzDpSharedNoMask;
typedef LzDpSharedT<LzDpRingLds<LZ_DP_MAXW, LzDpCells16>> LzDpShared16;
typedef LzDpSharedT<LzDpRingLdsNoMask<LZ_DP_MAXW, LzDpCells16>> LzDpSharedNoMask16;
typedef LzDpSharedT<LzDpRingHbm> LzDpSharedWide;

// Sweep state of one DP.  Only lane 0 reads and writes it, so it lives in that lane's registers:
// the row-end / row-set-up step is a serial piece of code that every other lane waits for, and with
// the state in LDS it was a chain of dependent LDS round trips.
struct LzDpCtl {
    s32 L, R; u32 LY, RY, prevLY, row;
    s32 best; u32 end1, end2;
    u32 tb_used, done, status, truncated;
    u32 b_hi;                             // columns < b_hi are staged in bb[]
    u32 max_row, min_col, max_col; u64 cells;
    // the bounds and the masks, from the job's pieces: the piece in force and its index (== the count: the bound is gone)
    u32 lbi, rbi; LzDpPiece lb, rb;
    u32 mk_lo, mk_hi;                     // mask pieces [mk_lo, mk_hi) may hold cells of the current row
    u32 mk_lo_r1, mk_next_r0;             // last row of piece mk_lo; first row of piece mk_hi (the next to come into reach)
};

struct LzDpLane {                       // per-lane values carried between the steps of one row (registers on the GPU)
    s32 c_left_old;
    s32 A, K; u32 cut; s32 i_in;        // walk-1 summary f(x) = cut ? A : max(A, x-K), and the scanned input
    s32 cand; u32 cand_col; s32 run_in; // best diagonal-won cell of the block; running best entering the block
    u32 first, last;                    // first / last live column of the block (0xFFFFFFFF: none)
    s32 bnd; u32 bnd_row, bnd_col, bnd_has;   // no_trim: the lane's best diagonal-won live cell on row M / column N, the latest on ties
    u32 tb_v;                           // traceback: the link this lane of the leading wave fetched for the window
    // the lane's first LZ_DP_BATCH cells, carried from walk to walk (with two columns per lane -- rows up to 512 wide --
    // that is the whole block: walk 2 and walk 3 then read nothing from the sweep row)
    s32 k_cc[LZ_DP_BATCH], k_dd[LZ_DP_BATCH], k_sc[LZ_DP_BATCH]; u32 k_mk[LZ_DP_BATCH], k_lk[LZ_DP_BATCH];
};
// Cross-lane steps are provided by the executor X (wave shuffles on the GPU, plain loops in the
// test harness); their semantics are fixed here:
//   X::scan_gap(sh,x0): r[l].i_in = (f_{l-1} o ... o f_0)(x0); returns (f_63 o ... o f_0)(x0)
//                       composition g o f: A = g.cut ? g.A : max(g.A, f.A - g.K), K = f.K + g.K, cut = f.cut | g.cut
//   X::scan_gap_plain(sh,x0,gapE,cpl,width): the same when no lane has cut set and every lane's K is gapE times its cells
//                       (lane l: min(cpl, max(0, width - l * cpl)) cells) -- an executor may use the closed form of sum K
//   X::scan_cand(sh,b0): r[l].run_in = max(b0, cand_0 .. cand_{l-1})
//   X::reduce_row(..) : first live column (lowest lane having one), last live column (highest lane),
//                       max cand and the column of the LAST lane attaining it; lane 0 fetches them in
//                       its next phase with X::row_result
//   X::phase(f) runs f on every lane and ends with a barrier; X::step(f) has no barrier: f may touch
//   only registers and the LDS cells of its own columns, and is followed by a cross-lane step.
//   X::leader(f) runs the serial piece f() once per DP and ends with a barrier.  On the GPU one whole
//   wave (the one X::lead_lane() is in) executes it in lockstep on identical data (every lane stores the same values), so that
//   with its inputs passed through X::uni (value of the first lane = a scalar register) the sweep
//   state LzDpCtl and the arithmetic on it stay in scalar registers and on the scalar ALU.
LZ_HD LzDpGap lz_dp_gap_compose(const LzDpGap& f, const LzDpGap& g)      // g after f
{
    LzDpGap h;
    const s32 t = f.A - g.K;
    h.A = g.cut ? g.A : (g.A > t ? g.A : t);
    h.K = f.K + g.K; h.cut = f.cut | g.cut;
    return h;
}
LZ_HD s32 lz_dp_gap_apply(const LzDpGap& f, s32 x) { const s32 t = x - f.K; return f.cut ? f.A : (f.A > t ? f.A : t); }

// ---- sequence access: A is the vertical (target) string, B the horizontal (query) string, both
// 1-based in DP coordinates (src/gapped_extend.c:2512-2533)
LZ_HD u32 lz_dp_a(const LzDpParams& P, const LzDpJob& J, u32 row)
{ return J.reversed ? P.tdp[(s64)J.anchor1 + 1 - (s64)row] : P.tdp[(s64)J.anchor1 + (s64)row]; }
LZ_HD u32 lz_dp_b(const LzDpParams& P, const LzDpJob& J, u32 col)
{ return J.reversed ? P.qdp[(s64)J.anchor2 + 1 - (s64)col] : P.qdp[(s64)J.anchor2 + (s64)col]; }

#if defined(__HIP_DEVICE_COMPILE__)
#define LZ_CLOCK() ((u64)__builtin_readcyclecounter())
#define LZ_REALTIME() ((u64)__builtin_amdgcn_s_memrealtime())
#else
#define LZ_CLOCK() ((u64)0)
#define LZ_REALTIME() ((u64)0)
#endif
// per-row step clocks (t_ph[], LZGPU_DPPROF): four s_memtime reads per row cost ~2-3 % of the sweep, so they
// are compiled in only with -DLZ_DP_PHASE_CLOCKS; the two per-DP totals are always taken
#if defined(LZ_DP_PHASE_CLOCKS)
#define LZ_PHASE_CLOCK() LZ_CLOCK()
#else
#define LZ_PHASE_CLOCK() ((u64)0)
#endif
#define LZ_SDIFF(a, b) (((s32)(a)) - ((s32)(b)))
// lane x cells-per-lane and the like: 24-bit operands, one full-rate instruction (a 32 x 32-bit v_mul_lo_u32 runs at a quarter of
// the rate, and the batch launches of k_ydrop are bound by vector-instruction issue)
#if defined(__HIP_DEVICE_COMPILE__)
LZ_HD u32 lz_mul24(u32 a, u32 b) { return __umul24(a, b); }
#else
LZ_HD u32 lz_mul24(u32 a, u32 b) { return a * b; }
#endif
#define LZ_RING(c) ((c) & (SH::RING - 1))

LZ_HD u32 lz_dp_special_min(u32 ry, s32 r) { if (r <= 0) return 0; if ((u32)r < ry) return (u32)r; return ry; }

// A value that lane-0 code keeps across rows must not stay "pending on a global load" in the eyes of the
// compiler: a later use would then wait for vmcnt(0), i.e. for every traceback store still in flight,
// on every row.  Passing it through X::uni right where it is loaded settles it there.
template <class X> LZ_HD LzDpPiece lz_dp_uni_piece(X& x, const LzDpPiece& g)
{ LzDpPiece r; r.r0 = x.uni(g.r0); r.r1 = x.uni(g.r1); r.x0 = x.uni(g.x0); r.fl = x.uni(g.fl); return r; }
LZ_HD s32 lz_dp_piece_at(const LzDpPiece& p, u32 row) { return p.x0 + ((p.fl & 1u) ? (s32)(row - p.r0) : 0); }

// ------------------------------------------------------------------------------------------------
// NOTRIM: !trimToPeak, a compile-time switch (its four per-lane values cost the default kernel nine spilled registers
// when it was a run-time flag)
// BOUNDS: the problem has earlier alignments (bounds to follow, segments to mask).  Without any -- the first round of a
// strand, where the longest DPs run -- the two routines of the row set-up, the mask stamps and their tests in the walks
// drop out at compile time (7.2 k -> 6.7 k cycles per row with the routines skipped by a run-time test alone).
// REPLICATE: see REPL below.
template <bool NOTRIM, bool BOUNDS, bool REPLICATE, class X, class SH>
LZ_HD void lz_dp_run(X& x, SH& sh, const LzDpParams& P, const LzDpJob& J,
                     const s32* tab /*[32*32] unmasked score classes*/, LzDpResult* res)
{
    const s32 gapE = P.gap_e, gapOE = P.gap_oe, Y = P.ydrop;
    const s32 RK = Y + gapOE + 1 + LZ_DP_ROW16_MARGIN;           // 16-bit sweep rows: base(row) = best at the start of the row - RK
    const u32 M = J.M, N = J.N;
    u8*  tb   = P.tb_arena  + J.tb_off;
    u32* trow = P.row_arena + J.row_off;
    u32* ops  = P.ops_arena + J.ops_off;
    const LzDpPiece* const pc_lb = P.pc_arena + J.pc_off;        // the job's pieces (BOUNDS): left bound, right bound, masks
    const LzDpPiece* const pc_rb = pc_lb + J.n_lb;
    const LzDpPiece* const pc_mk = pc_rb + J.n_rb;

    if (N == 0 || M == 0) {                                     // :3466-3467
        x.phase([&](int lane, LzDpLane&) {
            if (lane == 0) { res->score = 0; res->end1 = res->end2 = 0; res->n_ops = 0; res->status = LZ_DP_OK; res->truncated = 0;
                             res->max_row = res->min_col = res->max_col = 0; res->tb_used = 0; res->cells = 0; res->t_rows = res->t_trace = 0; res->t_begin = res->t_end = 0; for (int q = 0; q < 4; q++) res->t_ph[q] = 0; for (int q = 0; q < 5; q++) res->t_ld[q] = 0; } });
        return;
    }

    const u64 t0 = LZ_CLOCK();
    const u64 rt0 = LZ_REALTIME();
    // REPL: every wave keeps its own copy of the sweep state and runs the serial piece itself, on identical inputs.  One
    // wave doing it for all costs a barrier, sixteen words written to LDS and read back by the others, per row; the
    // copies cost nothing (the other three waves were waiting).  What the piece writes to LDS is the same from every
    // wave and each wave reads its own writes.  Global stores stay with one wave.  (Rounds 3-4 could replicate only the DPs without
    // bounds: the list of active segments was updated in place.  The pieces of round 5 are read-only: every wave keeps its own cursors.)
    // The copies are not free when a CU is full: a SIMD issues one scalar instruction per four cycles whichever wave it
    // comes from, and with seven DPs per CU every SIMD then carries seven copies of the piece instead of two (bench
    // pair, both strands in one launch of 4468 DPs: 72.2 ms against 70.4 with one leading wave; one strand's 2300 DPs,
    // whose launch lasts as long as its longest DP: 0.145 against 0.161 s for the two stages).  The launcher picks.
    constexpr bool REPL = REPLICATE;
    LzDpCtl ct;                                                 // lane 0's (REPL: every wave's)
    // ---- set-up + row 0 (:3500-3605)
    auto setup = [&]() {
        ct.L = 0; ct.R = (s32)N + 1;                           // (row 0 is not bounded: the first bound is read for row 1)
        ct.lbi = ct.rbi = 0; ct.mk_lo = ct.mk_hi = 0; ct.mk_lo_r1 = 0; ct.mk_next_r0 = 0xFFFFFFFFu;
        ct.lb.r0 = ct.lb.r1 = 0; ct.lb.x0 = 0; ct.lb.fl = 0; ct.rb = ct.lb;
        if (BOUNDS) {
            if (J.n_lb) ct.lb = lz_dp_uni_piece(x, pc_lb[0]);
            if (J.n_rb) ct.rb = lz_dp_uni_piece(x, pc_rb[0]);
            if (J.n_mk) { ct.mk_next_r0 = x.uni(pc_mk[0].r0); ct.mk_lo_r1 = x.uni(pc_mk[0].r1); }
        }
        ct.done = 0; ct.status = LZ_DP_OK; ct.truncated = 0;
        ct.best = 0; ct.end1 = ct.end2 = 0; ct.row = 0; ct.cells = 0;
        ct.max_row = 0; ct.min_col = 0; ct.max_col = 0;
        // row 0: C[0][0]=0, then insertions while the PREVIOUS column's C is >= -yDrop (note 13)
        u32 n0 = 1; s32 prevc = 0, c = -gapOE;
        while (n0 <= N && prevc >= -Y) { prevc = c; c -= gapE; n0++; }
        if (n0 + LZ_DP_LANES + 72 > SH::RING) { ct.status = LZ_DP_TOO_WIDE; ct.done = 1; }
        if (n0 > J.tb_cap || J.row_cap < 2) { ct.status = LZ_DP_TB_SLOT; ct.done = 1; }
        ct.LY = 0; ct.RY = n0; ct.tb_used = n0; ct.cells = n0;
        if (!ct.done && (!REPL || x.lead_here())) trow[0] = 0;
        ct.b_hi = 1; sh.trow_cur = 0;
        while (ct.RY + 2 > ct.b_hi) ct.b_hi += LZ_DP_LANES;     // columns [1, b_hi) are staged by the next phase
        ct.max_col = n0 ? n0 - 1 : 0;
        sh.done = ct.done; sh.b_hi = ct.b_hi; sh.ry_iter = n0; sh.row = 0; sh.LY = 0; sh.best = 0; sh.mk_lo = sh.mk_hi = 0; sh.extra = 0;
    };
    if (REPL) x.every_wave(setup); else x.leader(setup);
    if (!sh.done) {
        x.phase([&](int lane, LzDpLane&) {
            // the mask stamps are row numbers: a previous job's stamps must not survive in the LDS block
            if (BOUNDS) for (u32 k = (u32)lane; k < SH::RING; k += LZ_DP_LANES) sh.mk[k] = 0;
            for (u32 col = 1 + (u32)lane; col < sh.b_hi; col += LZ_DP_LANES) sh.bb[LZ_RING(col)] = (col <= N) ? (u8)(lz_dp_b(P, J, col) & 31u) : 0;
            sh.aa[lane] = (1 + (u32)lane <= M) ? (u8)(lz_dp_a(P, J, 1 + (u32)lane) & 31u) : 0;      // rows 1..64
            for (u32 col = (u32)lane; col < sh.ry_iter; col += LZ_DP_LANES) {
                s32 c = (col == 0) ? 0 : -gapOE - (s32)(col - 1) * gapE;
                sh.st_cd(LZ_RING(col), -RK, c, c - gapOE);      // (row 0: best = 0)
                tb[col] = (col == 0) ? 0 : LZ_C_FROM_I;
            }
        });
    }

    // ---- rows 1..M (:3607-3828).  One iteration = the row-end step of the row the previous iteration
    // swept, fused with the set-up of the next row (both are lane-0 work: one barrier instead of three),
    // then the three walks, separated only by the barrier inside each cross-lane step.
    if (NOTRIM) x.step([&](int, LzDpLane& r) { r.bnd = 0; r.bnd_row = r.bnd_col = 0; r.bnd_has = 0; });
    u32 row = 0, LY0 = 0, RYi = 0, cpl = 0, trow_cur = 0; s32 best0 = 0, i_last = 0;     // of the row in flight (uniform)
    bool swept = false;
    u64 tp0 = 0, tp1 = 0, tp2 = 0, tp3 = 0;
    u64 tl[5] = { 0, 0, 0, 0, 0 };
    // (whether the sweep is over is known from the read behind the serial piece: the loop test reads nothing)
    bool finished = REPL ? x.uni(ct.done) != 0u : sh.done != 0u;
    u32 arow_next = 0;                                          // the A class of the NEXT row, fetched a row ahead (behind walk 3's reads)
    while (!finished) {
        const u64 ts = LZ_PHASE_CLOCK();
        // (published at the end, in one block -- or, REPL, simply kept)
        u32 p_fill_n = 0, p_fill_base = 0, p_fill_trow = 0, p_stage_lo = ct.b_hi, p_stage_a = 0, p_trow_cur = 0, p_ry_iter = 0, p_cpl = 0, p_extra = 0;
        s32 p_fill_i = 0;
        auto control = [&]() {
            u64 q0 = LZ_PHASE_CLOCK(), q1 = q0, q2 = q0, q3 = q0, q4 = q0;
            [&]() {
            u32 extra = 0;
            if (swept) {
                // row end: new LY, best/end, right bound, overhang (:3769-3827)
                u32 first, last, ccol; s32 cmax;
                x.row_result(sh, first, last, cmax, ccol);
                first = x.uni(first); last = x.uni(last); cmax = x.uni(cmax); ccol = x.uni(ccol);
                q1 = LZ_PHASE_CLOCK();
                const u32 iter = RYi - LY0;
                ct.cells += iter;
                u32 tb_used = ct.tb_used + iter;
                s32 best = best0;
                if (cmax >= best0) { best = cmax; ct.best = cmax; ct.end1 = row; ct.end2 = ccol; }      // :3731-3735
                if (first == 0xFFFFFFFFu) { ct.tb_used = tb_used; ct.LY = RYi; ct.done = 1; return; }   // LY >= RY: feasible region empty
                ct.LY = first;
                if (LY0 < ct.min_col) ct.min_col = LY0;
                ct.max_row = row;
                // (:3787 asks for the RIGHT neighbour's segment whichever way the sweep goes; in a backward sweep that segment drives the
                // left bound -- lzh_dp_pieces -- so "it is still there" is read off the left bound's pieces then)
                const bool right_seg_alive = BOUNDS && (J.reversed ? ct.lbi < J.n_lb : ct.rbi < J.n_rb);
                const s32 NN = (right_seg_alive && ct.R > 0) ? ct.R - 1 : (s32)N;
                u32 RY = RYi, np = 0;
                if (RY > last + 1) RY = last + 1;
                else {
                    const s32 thr = best - Y; s32 i = i_last;
                    while (i >= thr && (s32)RY <= NN) { np++; RY++; i -= gapE; }
                }
                // overhang cells C=i, D=i-gapOE, link=I (:3799-3811): a few are stored here, a long run by all lanes
                if (np <= LZ_DP_SERIAL_FILL) {
                    const u32 base = RY - np;
                    for (u32 k = 0; k < np; k++) {
                        const s32 iv = i_last - (s32)k * gapE;
                        sh.st_cd(LZ_RING(base + k), best0 - RK, iv, iv - gapOE);     // (cells of the row just swept: its base)
                        if (!REPL || x.lead_here()) tb[(u32)(trow_cur + base + k)] = LZ_C_FROM_I;
                    }
                } else { p_fill_n = np; p_fill_base = RY - np; p_fill_i = i_last; p_fill_trow = trow_cur; extra = 1; }
                tb_used += np;
                if (RY - 1 > ct.max_col) ct.max_col = RY - 1;
                if ((s32)RY <= NN) { sh.st_dead(LZ_RING(RY)); RY++; }   // terminating cell, :3818-3826
                ct.RY = RY; ct.tb_used = tb_used;
                // B classes of the columns the next row may reach; every LZ_DP_LANES rows the next block of A classes
                u32 bh = ct.b_hi;
                p_stage_lo = bh;
                while (RY + 2 > bh) { bh += LZ_DP_LANES; extra = 1; }
                ct.b_hi = bh;
                p_stage_a = ((row & (LZ_DP_LANES - 1)) == 0) ? row + 1 : 0;
                if (p_stage_a) extra = 1;
            }
            p_extra = extra;
            q2 = LZ_PHASE_CLOCK();
            // set-up of the next row: bounds, active segments, traceback budget
            if (ct.row >= M) { ct.done = 1; return; }
            ct.row++;
            ct.prevLY = ct.LY;
            if (BOUNDS) {
                if (ct.row > J.horizon) { ct.status = LZ_DP_PIECE_SLOT; ct.done = 1; return; }      // the pieces end here: run again with more of them
                // the row's bounds (update_LR_bounds, :4588-4700): the piece in force, or the next one when the row has passed its end
                // (a bound that ends leaves 0 + 1 / 0 - 1 behind, as the reference's last look for a next segment does: R is still read by
                // the row end of a backward sweep, :3787, after the segment that drove it is gone)
                while (ct.lbi < J.n_lb && ct.row > ct.lb.r1) { ct.lbi++; if (ct.lbi < J.n_lb) ct.lb = lz_dp_uni_piece(x, pc_lb[ct.lbi]); else ct.L = 1; }
                if (ct.lbi < J.n_lb) { ct.L = lz_dp_piece_at(ct.lb, ct.row); ct.LY = (u32)(((s32)ct.LY > ct.L) ? (s32)ct.LY : ct.L); }
                while (ct.rbi < J.n_rb && ct.row > ct.rb.r1) { ct.rbi++; if (ct.rbi < J.n_rb) ct.rb = lz_dp_uni_piece(x, pc_rb[ct.rbi]); else ct.R = -1; }
                if (ct.rbi < J.n_rb) { ct.R = lz_dp_piece_at(ct.rb, ct.row); ct.RY = lz_dp_special_min(ct.RY, ct.R); }
            }
            q3 = LZ_PHASE_CLOCK();
            if (BOUNDS) {
                // the mask pieces in reach of the row (update_active_segs, :4885-4965): those that have begun, from the first that has not ended
                while (ct.mk_hi < J.n_mk && ct.mk_next_r0 <= ct.row) { ct.mk_hi++; if (ct.mk_hi < J.n_mk) ct.mk_next_r0 = x.uni(pc_mk[ct.mk_hi].r0); }
                while (ct.mk_lo < ct.mk_hi && ct.mk_lo_r1 < ct.row) { ct.mk_lo++; if (ct.mk_lo < J.n_mk) ct.mk_lo_r1 = x.uni(pc_mk[ct.mk_lo].r1); }
            }
            q4 = LZ_PHASE_CLOCK();
            if (ct.RY < ct.LY) ct.RY = ct.LY;                   // note 11
            const u32 width = ct.RY - ct.LY;
            const s32 tb_needed = (s32)width + P.ydrop_tail;
            if ((s64)ct.tb_used + tb_needed >= (s64)P.tb_len) { ct.truncated = 1; ct.done = 1; ct.row--; return; }   // :3640-3661
            if ((u64)ct.tb_used + (u64)tb_needed > (u64)J.tb_cap) { ct.status = LZ_DP_TB_SLOT; ct.done = 1; return; }
            if (width + (u32)P.ydrop_tail + LZ_DP_LANES + 72 > SH::RING) { ct.status = LZ_DP_TOO_WIDE; ct.done = 1; return; }
            if (ct.row + 1 >= J.row_cap) { ct.status = LZ_DP_ROW_SLOT; ct.done = 1; return; }
            p_trow_cur = ct.tb_used - ct.LY;                    // tbRow[row], :3662 (u32 wrap intended)
            if (!REPL || x.lead_here()) trow[ct.row] = p_trow_cur;
            p_ry_iter = ct.RY;
            p_cpl = (width + LZ_DP_LANES - 1) / LZ_DP_LANES;
            }();
            if (!REPL) {
                sh.row = ct.row; sh.LY = ct.LY; sh.ry_iter = p_ry_iter; sh.cpl = p_cpl;
                sh.best = ct.best; sh.trow_cur = p_trow_cur; sh.done = ct.done; sh.extra = p_extra;
                sh.mk_lo = ct.mk_lo; sh.mk_hi = ct.mk_hi; sh.fill_n = p_fill_n; sh.fill_base = p_fill_base; sh.fill_trow = p_fill_trow;
                sh.stage_lo = p_stage_lo; sh.stage_a = p_stage_a; sh.fill_i = p_fill_i; sh.b_hi = ct.b_hi;
            }
            const u64 q5 = LZ_PHASE_CLOCK();
            if (q1 < q0) q1 = q0; if (q2 < q1) q2 = q1; if (q3 < q2) q3 = q2; if (q4 < q3) q4 = q3;
            tl[0] += q1 - q0; tl[1] += q2 - q1; tl[2] += q3 - q2; tl[3] += q4 - q3; tl[4] += q5 - q4;
        };
        if (REPL) x.every_wave(control); else x.leader(control);
        u32 e_on, mk_lo_now = 0, mk_hi_now = 0;
        const s32 base_prev = best0 - RK;                        // base of the row just swept (best0 is still that row's)
        if (REPL) {
            finished = x.uni(ct.done) != 0u; e_on = p_extra;
            if (BOUNDS) { mk_lo_now = x.uni(ct.mk_lo); mk_hi_now = x.uni(ct.mk_hi); }
            row = x.uni(ct.row); LY0 = x.uni(ct.LY); RYi = x.uni(p_ry_iter); cpl = x.uni(p_cpl); best0 = x.uni(ct.best); trow_cur = x.uni(p_trow_cur);
        } else {
            // the published words of every row, read together (two 128-bit reads and one wait), then the branches
            const u32 v_row = sh.row, v_ly = sh.LY, v_ry = sh.ry_iter, v_cpl = sh.cpl, v_trow = sh.trow_cur, v_done = sh.done, v_extra = sh.extra;
            const s32 v_best = sh.best;
            if (BOUNDS) { mk_lo_now = x.uni((u32)sh.mk_lo); mk_hi_now = x.uni((u32)sh.mk_hi); }
            finished = x.uni(v_done) != 0u; e_on = x.uni(v_extra);
            row = x.uni(v_row); LY0 = x.uni(v_ly); RYi = x.uni(v_ry); cpl = x.uni(v_cpl); best0 = x.uni(v_best); trow_cur = x.uni(v_trow);
        }
        if (finished) break;
        const s32 base_cur = best0 - RK;                         // ... and of the row about to be
        // the rare parallel pieces of the row set-up: a long run of overhang cells, the next columns' / rows' classes
        if (e_on) {
            const u32 e_fill_n = REPL ? p_fill_n : sh.fill_n, e_fill_base = REPL ? p_fill_base : sh.fill_base, e_fill_trow = REPL ? p_fill_trow : sh.fill_trow;
            const s32 e_fill_i = REPL ? p_fill_i : sh.fill_i;
            const u32 e_stage_lo = REPL ? p_stage_lo : sh.stage_lo, e_b_hi = REPL ? ct.b_hi : sh.b_hi, e_stage_a = REPL ? p_stage_a : sh.stage_a;
            x.phase([&](int lane, LzDpLane&) {
                for (u32 k = (u32)lane; k < e_fill_n; k += LZ_DP_LANES) {
                    const s32 iv = e_fill_i - (s32)k * gapE;
                    sh.st_cd(LZ_RING(e_fill_base + k), base_prev, iv, iv - gapOE);
                    tb[(u32)(e_fill_trow + e_fill_base + k)] = LZ_C_FROM_I;
                }
                for (u32 col = e_stage_lo + (u32)lane; col < e_b_hi; col += LZ_DP_LANES)
                    sh.bb[LZ_RING(col)] = (col <= N) ? (u8)(lz_dp_b(P, J, col) & 31u) : 0;
                if (e_stage_a) {
                    const u32 r2 = e_stage_a + (u32)lane;
                    sh.aa[lane] = (r2 <= M) ? (u8)(lz_dp_a(P, J, r2) & 31u) : 0;
                }
            });
        }
        swept = true;
        const bool any_active = BOUNDS && mk_hi_now > mk_lo_now;
        const u32 row_stamp = SH::stamp(row);
        // the 16-bit stamps start over: none of the old ones may survive (all lanes, behind the previous row's last barrier; its own barrier
        // before this row's stamps are written)
        if (BOUNDS && SH::STAMP_WRAPS && row > 1 && row_stamp == 1u)
            x.phase([&](int lane, LzDpLane&) { for (u32 k = (u32)lane; k < SH::RING; k += LZ_DP_LANES) sh.mk[k] = 0; });
        // The row's masked cells, stamped by the lanes: lane l takes pieces mk_lo + l, mk_lo + l + LANES, ... of those in reach; a piece
        // that covers the row masks x .. x + extra, clipped to the band (that also keeps the ring free of aliases: RY - LY < RING).
        // (build_active_seg, :4992-5035; the reference walks a linked list of segments here, one at a time.)
        if (any_active) {
            x.phase([&](int lane, LzDpLane&) {
                for (u32 k = mk_lo_now + (u32)lane; k < mk_hi_now; k += LZ_DP_LANES) {
                    const LzDpPiece pc = pc_mk[k];
                    if (row < pc.r0 || row > pc.r1) continue;
                    const s32 xs = lz_dp_piece_at(pc, row);
                    if (xs < 0) continue;                                  // (the reference keeps the column unsigned: left of column 0 it is outside every band)
                    const u32 x0 = (u32)xs, x1 = x0 + (pc.fl >> 1);
                    const u32 lo = x0 > LY0 ? x0 : LY0, hi = x1 < RYi ? x1 : RYi;
                    for (u32 c = lo; c <= hi; c++) sh.mk[LZ_RING(c)] = (typename SH::stamp_t)row_stamp;
                }
            });
        }
        // the row's A class: fetched during the previous row (arow_next) unless this row opens a freshly staged block of aa[]
        const u32 aidx = (row - 1) & (LZ_DP_LANES - 1);
        const u32 arow = aidx == 0 ? x.uni((u32)sh.aa[0]) : arow_next;
        const s32* trow_tab = tab + (arow << 5);

        // The walks read the sweep row in batches of LZ_DP_BATCH cells: all LDS reads of a batch are
        // issued before the serial recurrence consumes them (one wave per SIMD has nothing else to
        // hide LDS latency behind).  A lane's FIRST batch is read once, by walk 1, and carried in registers
        // (LzDpLane::k_*) through walks 2 and 3.
        auto load_batch = [&](u32 base, s32* vcc, s32* vdd, s32* vsc, u32* vmk) {
            u32 vbb[LZ_DP_BATCH];
            LZ_UNROLL for (int k = 0; k < LZ_DP_BATCH; k++) {
                const u32 rx = LZ_RING(base + k);
                sh.ld_cd(rx, base_prev, vcc[k], vdd[k]); if (BOUNDS && any_active) vmk[k] = sh.mk[rx]; else vmk[k] = 0u; vbb[k] = sh.bb[rx];   // (any_active is uniform: no stamp is read on a row without a mask piece in reach)
            }
            LZ_UNROLL for (int k = 0; k < LZ_DP_BATCH; k++) vsc[k] = trow_tab[vbb[k] & 31u];
        };
        // walk 1: block summaries of the insertion recurrence
        const u64 ta = LZ_PHASE_CLOCK();
        x.step([&](int lane, LzDpLane& r) {
            u32 c0 = LY0 + lz_mul24((u32)lane, cpl), c1 = c0 + cpl; if (c1 > RYi) c1 = RYi;
            s32 A = LZ_DP_NEGINF - (1 << 24), K = 0; u32 cut = 0;
            const s32 cl = sh.ld_c(LZ_RING(c0 - 1u), base_prev);             // (unconditional: issued with the batch's reads; selected below)
            load_batch(c0, r.k_cc, r.k_dd, r.k_sc, r.k_mk);
            r.c_left_old = (c0 < RYi && c0 > LY0) ? cl : LZ_DP_NEGINF;
            s32 c_left = r.c_left_old;
            // (column LY0 -- no cell to its left, :3697 -- can only be the first cell of the lane whose block starts the row: the test is made for that
            // cell alone, FIRST = the lane's first batch)
            const bool at_ly = c0 == LY0;
            auto cells = [&](auto first_tag, u32 base, const s32* vcc, const s32* vdd, const s32* vsc, const u32* vmk) {
                constexpr bool FIRST = decltype(first_tag)::value;
                LZ_UNROLL for (int k = 0; k < LZ_DP_BATCH; k++) {
                    const u32 col = base + k;
                    if (col < c1) {
                        const s32 cin = (FIRST && k == 0 && at_ly) ? LZ_DP_NEGINF : c_left + vsc[k];
                        const s32 d = vdd[k];
                        c_left = vcc[k];
                        const bool masked = any_active && vmk[k] == row_stamp;
                        if (masked) { A = LZ_DP_NEGINF; K = 0; cut = 1; }
                        else {
                            const s32 a2 = A - gapE;
                            const s32 open = (d > cin) ? (LZ_DP_NEGINF - (1 << 24)) : cin - gapOE;
                            A = open > a2 ? open : a2;
                            K += gapE;
                        }
                    }
                }
            };
            cells(std::true_type(), c0, r.k_cc, r.k_dd, r.k_sc, r.k_mk);
            for (u32 base = c0 + LZ_DP_BATCH; base < c1; base += LZ_DP_BATCH) {
                s32 vcc[LZ_DP_BATCH], vdd[LZ_DP_BATCH], vsc[LZ_DP_BATCH]; u32 vmk[LZ_DP_BATCH];
                load_batch(base, vcc, vdd, vsc, vmk);
                cells(std::false_type(), base, vcc, vdd, vsc, vmk);
            }
            r.A = A; r.K = K; r.cut = cut;
        });
        // 64-lane exclusive scan of the block summaries, x0 = -inf (:3679 "i = negInf")
        // (a row without a masked cell -- every row of a DP without bounds, and the rows of the others on which no mask piece is in reach --
        // has cut == 0 and K = gapE per cell in every lane: the closed form of the scan; any_active is uniform)
        if (BOUNDS && any_active) i_last = x.uni(x.scan_gap(sh, LZ_DP_NEGINF));
        else                      i_last = x.uni(x.scan_gap_plain(sh, LZ_DP_NEGINF, gapE, cpl, RYi - LY0));
        const u64 tb_ = LZ_PHASE_CLOCK();
        // walk 2: the cells (:3697-3767 without the prune test), candidate bests
        x.step([&](int lane, LzDpLane& r) {
            u32 c0 = LY0 + lz_mul24((u32)lane, cpl), c1 = c0 + cpl; if (c1 > RYi) c1 = RYi;
            s32 i = r.i_in, c_left = r.c_left_old;
            s32 cmax = LZ_DP_NEGINF - (1 << 24); u32 ccol = 0;
            // one batch: the new C, D and links replace the old values in vcc / vdd / vlk
            const bool at_ly = c0 == LY0;
            auto cells = [&](auto first_tag, u32 base, s32* vcc, s32* vdd, const s32* vsc, const u32* vmk, u32* vlk) {
                constexpr bool FIRST = decltype(first_tag)::value;
                LZ_UNROLL for (int k = 0; k < LZ_DP_BATCH; k++) {
                    const u32 col = base + k;
                    vlk[k] = 0;
                    if (col < c1) {
                        s32 c = (FIRST && k == 0 && at_ly) ? LZ_DP_NEGINF : c_left + vsc[k];
                        s32 d = vdd[k];
                        c_left = vcc[k];
                        const bool masked = any_active && vmk[k] == row_stamp;
                        u32 link;
                        if (masked) { link = 0x80; c = LZ_DP_NEGINF; d = LZ_DP_NEGINF; i = LZ_DP_NEGINF; }
                        else {
                            // The two cases of :3708-3767 -- a gap wins the cell (d > c || i > c: C = the larger gap, both gaps extend) or the
                            // diagonal does (candidate best; each gap either opens from C or extends) -- as selects on one instruction stream:
                            // neighbouring lanes take different cases on most rows, and as two branches each lane paid for both plus the copies
                            // that merge them.
                            const s32 m = d > i ? d : i;
                            const bool gap = m > c, from_d = d >= i;
                            const s32 dn = d - gapE, in = i - gapE, c_open = c - gapOE;
                            const bool open_d = !gap && c_open > dn, open_i = !gap && c_open > in;
                            const bool cand = !gap && c >= cmax;             // candidate for bestScore (later column wins ties)
                            cmax = cand ? c : cmax; ccol = cand ? col : ccol;
                            link = gap ? (from_d ? (u32)(LZ_C_FROM_D | LZ_I_EXT | LZ_D_EXT) : (u32)(LZ_C_FROM_I | LZ_I_EXT | LZ_D_EXT))
                                       : ((open_d ? 0u : (u32)LZ_D_EXT) | (open_i ? 0u : (u32)LZ_I_EXT) | (u32)LZ_C_FROM_C);
                            c = gap ? m : c;
                            d = open_d ? c_open : dn;
                            i = open_i ? c_open : in;
                        }
                        if (i < LZ_DP_NEGINF - (1 << 24)) i = LZ_DP_NEGINF - (1 << 24);
                        vcc[k] = c; vdd[k] = d; vlk[k] = link;
                    }
                }
                LZ_UNROLL for (int k = 0; k < LZ_DP_BATCH; k++) {
                    const u32 col = base + k;
                    if (col < c1) { const u32 rx = LZ_RING(col); sh.st_cd(rx, base_cur, vcc[k], vdd[k]); if (!FIRST) sh.lk[rx] = (u8)vlk[k]; }   // (walk 3 has the first batch's links in registers)
                }
            };
            cells(std::true_type(), c0, r.k_cc, r.k_dd, r.k_sc, r.k_mk, r.k_lk);  // (k_cc / k_lk go on to walk 3)
            for (u32 base = c0 + LZ_DP_BATCH; base < c1; base += LZ_DP_BATCH) {
                s32 vcc[LZ_DP_BATCH], vdd[LZ_DP_BATCH], vsc[LZ_DP_BATCH]; u32 vmk[LZ_DP_BATCH], vlk[LZ_DP_BATCH];
                load_batch(base, vcc, vdd, vsc, vmk);
                cells(std::false_type(), base, vcc, vdd, vsc, vmk, vlk);
            }
            r.cand = cmax; r.cand_col = ccol;
        });
        // 64-lane exclusive prefix max of the candidates, seeded with bestScore at row start
        x.scan_cand(sh, best0);
        const u64 tc = LZ_PHASE_CLOCK();
        // walk 3: prune test against the running best, final stores, traceback bytes
        const u32 a_nx = sh.aa[row & (LZ_DP_LANES - 1)];        // (next row's A class: not used when that row opens a new block)
        x.step([&](int lane, LzDpLane& r) {
            u32 c0 = LY0 + lz_mul24((u32)lane, cpl), c1 = c0 + cpl; if (c1 > RYi) c1 = RYi;
            s32 rb = r.run_in;
            u32 first = 0xFFFFFFFFu, last = 0xFFFFFFFFu;
            u8* tbr = tb + (u32)(trow_cur + c0);
            auto cells = [&](u32 base, const s32* vcc, const u32* vlk) {
                LZ_UNROLL for (int k = 0; k < LZ_DP_BATCH; k++) {
                    const u32 col = base + k;
                    if (col < c1) {
                        const s32 c = vcc[k];
                        const u32 link = vlk[k];
                        const bool live = (link != 0x80) && (c >= rb - Y);
                        if (live) {
                            first = first < col ? first : col;      // (columns only grow: the first live one is the least; none yet = 0xFFFFFFFF)
                            last = col;
                            if ((link & 3u) == LZ_C_FROM_C && c > rb) rb = c;
                            // boundaryScore (:3747-3750) never feeds back into the sweep: every lane keeps its own
                            // maximum (>=: the later cell wins a tie) and the lanes are compared once, after the last row
                            if (NOTRIM && (link & 3u) == LZ_C_FROM_C && (row == M || col == N) && (!r.bnd_has || c >= r.bnd))
                                { r.bnd = c; r.bnd_row = row; r.bnd_col = col; r.bnd_has = 1; }
                            tbr[k] = (u8)link;
                        } else {
                            sh.st_dead(LZ_RING(col));
                            tbr[k] = 0;
                        }
                    }
                }
                tbr += LZ_DP_BATCH;
            };
            cells(c0, r.k_cc, r.k_lk);
            for (u32 base = c0 + LZ_DP_BATCH; base < c1; base += LZ_DP_BATCH) {
                s32 vcc[LZ_DP_BATCH]; u32 vlk[LZ_DP_BATCH];
                LZ_UNROLL for (int k = 0; k < LZ_DP_BATCH; k++) { const u32 rx = LZ_RING(base + k); vcc[k] = sh.ld_c(rx, base_cur); vlk[k] = sh.lk[rx]; }
                cells(base, vcc, vlk);
            }
            r.first = first; r.last = last;
        });
        x.reduce_row(sh);
        arow_next = x.uni(a_nx);
        const u64 td = LZ_PHASE_CLOCK();
        tp0 += ta - ts; tp1 += tb_ - ta; tp2 += tc - tb_; tp3 += td - tc;
    }

    // ---- traceback (:3847-3859).  The walk is one dependent byte per step; to keep it off HBM latency
    // the first 64 lanes fetch the links along the diagonal below the current point (the path of an
    // alignment is a diagonal between gaps), lane 0 then consumes the window from LDS for as long as the
    // path stays on that diagonal and asks for a new window after a gap.  Edit ops are run-length merged
    // in registers (edit_script_add, src/edit_script.c:261-300) and stored once per run.
    const u64 t1 = LZ_CLOCK();
    if (NOTRIM) {
        // endIsBoundary (:3744, :3750, :3866): bestScore's last update is the LAST cell attaining the maximum, (end1, end2);
        // boundaryScore's last update is the last boundary cell attaining the boundary maximum; whichever of the two
        // events came later in row-major order stands (the same cell: the boundary, it is tested second)
        x.phase([&](int lane, LzDpLane& r) {                    // (the sweep row is dead: its first cells carry the lanes' values)
            sh.scratch((u32)lane) = r.bnd_has ? r.bnd : LZ_DP_NEGINF - (1 << 24);
            sh.scratch(LZ_DP_LANES + (u32)lane) = (s32)r.bnd_row; sh.scratch(2 * LZ_DP_LANES + (u32)lane) = (s32)r.bnd_col;
            sh.scratch(3 * LZ_DP_LANES + (u32)lane) = (s32)r.bnd_has;
        });
        x.phase([&](int lane, LzDpLane&) {
            if (lane != x.lead_lane() || ct.status != LZ_DP_OK) return;
            bool has = false; s32 bc = 0; u32 br = 0, bcol = 0;
            for (int l = 0; l < LZ_DP_LANES; l++) {
                if (!sh.scratch(3 * LZ_DP_LANES + (u32)l)) continue;
                const s32 c = sh.scratch((u32)l); const u32 rr = (u32)sh.scratch(LZ_DP_LANES + (u32)l), cl = (u32)sh.scratch(2 * LZ_DP_LANES + (u32)l);
                if (!has || c > bc || (c == bc && (rr > br || (rr == br && cl > bcol)))) { has = true; bc = c; br = rr; bcol = cl; }
            }
            if (has && (br > ct.end1 || (br == ct.end1 && bcol >= ct.end2))) { ct.best = bc; ct.end1 = br; ct.end2 = bcol; }
        });
    }
    x.phase([&](int lane, LzDpLane&) {
        if (lane != x.lead_lane()) return;
        sh.tb_row = ct.end1; sh.tb_col = ct.end2; sh.tb_prev = 0; sh.tb_nops = 0; sh.tb_run_op = 0; sh.tb_run_len = 0;
        sh.tb_done = (ct.status != LZ_DP_OK) || !(ct.end1 >= 1 || ct.end2 > 0);
    });
    while (!sh.tb_done) {
        // the 64 links along the diagonal below the current point, one per lane of the leading wave, kept in registers
        x.phase([&](int lane, LzDpLane& r) {
            if (!x.in_lead_wave(lane)) return;
            const u32 k = (u32)lane & 63u;
            u32 v = 0xFFu;
            if (k <= sh.tb_row && k <= sh.tb_col) v = tb[(u32)(trow[sh.tb_row - k] + (sh.tb_col - k))];
            r.tb_v = v;
        });
        // The leading wave walks the window, all its lanes in lockstep on the same (scalar) state.  A run of diagonal
        // links is taken in ONE step: once the previous op is diagonal a link's op is its low two bits, so the run's
        // length is a count of trailing ones in a ballot -- the walk used to be one dependent LDS read and a dozen
        // dependent instructions per link, on one lane (350 cycles per link; 8 % of a long DP's time).
        x.leader([&]() {
            u32 row = x.uni(sh.tb_row), col = x.uni(sh.tb_col), prev_op = x.uni(sh.tb_prev), n_ops = x.uni(sh.tb_nops);
            u32 run_op = x.uni(sh.tb_run_op), run_len = x.uni(sh.tb_run_len), status = ct.status;
            const u64 diag = x.tb_ballot_diag();                // bit k: (link k & 3) is neither C_FROM_I nor C_FROM_D
            u32 k = 0;
            while (k < LZ_DP_TBWIN && (row >= 1 || col > 0)) {
                const u32 link = x.tb_link(k);
                u32 op = link & 3u;
                if (prev_op == LZ_C_FROM_I && (link & LZ_I_EXT)) op = LZ_C_FROM_I;
                if (prev_op == LZ_C_FROM_D && (link & LZ_D_EXT)) op = LZ_C_FROM_D;
                const u32 eop = (op == LZ_C_FROM_I) ? 1u : (op == LZ_C_FROM_D) ? 2u : 3u;
                u32 steps = 1;
                if (eop == 3u) {                                // the links that follow, for as long as they are diagonal too
                    const u64 rest = (k + 1u < 64u) ? (diag >> (k + 1u)) : 0ull;
                    u32 more = (u32)__builtin_ctzll(~rest);     // (rest has zeros on top: never all ones)
                    if (more > LZ_DP_TBWIN - 1u - k) more = LZ_DP_TBWIN - 1u - k;
                    const u32 lim = row > col ? row : col;      // step j of the run is taken while j < max(row, col): the walk ends at the origin
                    steps = 1u + more; if (steps > lim) steps = lim;
                }
                if (eop == run_op) run_len += steps;
                else {
                    if (run_len) { if (n_ops >= J.ops_cap) { status = LZ_DP_OPS_SLOT; break; } ops[n_ops++] = run_op | (run_len << 2); }
                    run_op = eop; run_len = steps;
                }
                prev_op = op;
                if (op == LZ_C_FROM_I)      { col--; break; }   // a gap: the window is stale
                else if (op == LZ_C_FROM_D) { row--; break; }
                row -= steps; col -= steps; k += steps;         // still on the window's diagonal
            }
            const bool fin = (status != LZ_DP_OK) || !(row >= 1 || col > 0);
            if (fin && status == LZ_DP_OK && run_len) {
                if (n_ops >= J.ops_cap) status = LZ_DP_OPS_SLOT; else ops[n_ops++] = run_op | (run_len << 2);
                run_len = 0;
            }
            sh.tb_row = row; sh.tb_col = col; sh.tb_prev = prev_op; sh.tb_nops = n_ops;
            sh.tb_run_op = run_op; sh.tb_run_len = run_len; ct.status = status; sh.tb_done = fin;
        });
    }
    x.phase([&](int lane, LzDpLane&) {
        if (lane != x.lead_lane()) return;
        res->score = ct.best; res->end1 = ct.end1; res->end2 = ct.end2; res->n_ops = sh.tb_nops;
        res->status = ct.status; res->truncated = ct.truncated;
        res->max_row = ct.max_row; res->min_col = ct.min_col; res->max_col = ct.max_col;
        res->tb_used = ct.tb_used; res->cells = ct.cells;
        res->t_rows = t1 - t0; res->t_trace = LZ_CLOCK() - t1; res->t_begin = rt0; res->t_end = LZ_REALTIME();
        res->t_ph[0] = tp0; res->t_ph[1] = tp1; res->t_ph[2] = tp2; res->t_ph[3] = tp3;
        for (int q = 0; q < 5; q++) res->t_ld[q] = tl[q];
    });
}
