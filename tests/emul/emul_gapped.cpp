// emul_gapped.cpp -- TEST INFRASTRUCTURE: executes the product's one-sided DP (lz_dp_dev.hpp, the
// code the gfx950 kernel runs, one wave per DP) lane by lane, phase by phase on the CPU, under the
// product's host orchestration (lz_gapped_host.cpp).  Lets "-m 'not gpu'" tests check the parallel
// row algorithm and the speculative anchor windows against the oracle.  Never part of liblzgpu.so.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "../../lastz_amd/csrc/lz_gapped_host.hpp"
#include "../../lastz_amd/csrc/lz_host.hpp"

struct CpuPhases {                      // X for lz_dp_run: a phase = the lambda for lanes 0..63
    LzDpLane lanes[LZ_DP_LANES];
    template <class F> void phase(F&& f) { for (int l = 0; l < LZ_DP_LANES; l++) f(l, lanes[l]); }
    template <class F> void step(F&& f)  { for (int l = LZ_DP_LANES - 1; l >= 0; l--) f(l, lanes[l]); }   // no barrier on the GPU: any lane order must do
    template <class F> void leader(F&& f) { f(); }
    bool in_lead_wave(int lane) const { return lane < 64; }
    u64 tb_ballot_diag() const { u64 m = 0; for (int l = 0; l < 64; l++) { const u32 o = lanes[l].tb_v & 3u; if (o != (u32)LZ_C_FROM_I && o != (u32)LZ_C_FROM_D) m |= 1ull << l; } return m; }
    u32 tb_link(u32 k) const { return lanes[k].tb_v; }
    template <class F> void every_wave(F&& f) { f(); }          // (one copy of the control state here)
    bool lead_here() const { return true; }
    int lead_lane() const { return 0; }
    s32 uni(s32 v) { return v; }
    u32 uni(u32 v) { return v; }
    void row_result(const LzDpSharedBase& sh, u32& first, u32& last, s32& cmax, u32& ccol) { first = sh.r_first; last = sh.r_last; cmax = sh.r_cmax; ccol = sh.r_ccol; }
    s32 scan_gap(LzDpSharedBase&, s32 x0) {
        s32 x = x0;
        for (int l = 0; l < LZ_DP_LANES; l++) { lanes[l].i_in = x; LzDpGap f = { lanes[l].A, lanes[l].K, lanes[l].cut }; x = lz_dp_gap_apply(f, x); }
        return x;
    }
    s32 scan_gap_plain(LzDpSharedBase& sh, s32 x0, s32 gap_e, u32 cpl, u32 width) {   // (checks what the GPU executor's closed form assumes)
        for (int l = 0; l < LZ_DP_LANES; l++) {
            const u32 lo = (u32)l * cpl, n = lo < width ? (width - lo < cpl ? width - lo : cpl) : 0u;
            if (lanes[l].cut != 0 || lanes[l].K != gap_e * (s32)n) { fprintf(stderr, "emul: scan_gap_plain's precondition does not hold (lane %d)\n", l); abort(); }
        }
        // the closed form, lane by lane, against the map algebra
        s32 pm = x0, x = x0;
        for (int l = 0; l < LZ_DP_LANES; l++) {
            const u32 ce = (u32)l * cpl, ci = ce + cpl;
            const s32 cum_excl = gap_e * (s32)(ce < width ? ce : width), cum_incl = gap_e * (s32)(ci < width ? ci : width);
            if (pm - cum_excl != x) { fprintf(stderr, "emul: closed form of the gap scan differs at lane %d\n", l); abort(); }
            lanes[l].i_in = x;
            LzDpGap f = { lanes[l].A, lanes[l].K, lanes[l].cut }; x = lz_dp_gap_apply(f, x);
            const s32 b = lanes[l].A + cum_incl; if (b > pm) pm = b;
        }
        if (pm - gap_e * (s32)width != x) { fprintf(stderr, "emul: closed form of the gap scan differs at the end\n"); abort(); }
        (void)sh;
        return x;
    }
    void scan_cand(LzDpSharedBase&, s32 b0) { s32 rb = b0; for (int l = 0; l < LZ_DP_LANES; l++) { lanes[l].run_in = rb; if (lanes[l].cand > rb) rb = lanes[l].cand; } }
    void reduce_row(LzDpSharedBase& sh) {
        u32 first = 0xFFFFFFFFu, last = 0xFFFFFFFFu, ccol = 0; s32 cmax = LZ_DP_NEGINF - (1 << 24);
        for (int l = 0; l < LZ_DP_LANES; l++) {
            if (lanes[l].first != 0xFFFFFFFFu) { if (first == 0xFFFFFFFFu) first = lanes[l].first; last = lanes[l].last; }
            if (lanes[l].cand >= cmax) { cmax = lanes[l].cand; ccol = lanes[l].cand_col; }
        }
        sh.r_first = first; sh.r_last = last; sh.r_cmax = cmax; sh.r_ccol = ccol;
    }
};

u32 emul_verify_pieces(const LzHostSnapshot& S, const LzDpJob& J, u32 horizon, u32 rows);   // emul_bounds_plain.cpp
static u64 g_pieces_verified = 0;
extern "C" u64 emul_gapped_pieces_verified(void) { return g_pieces_verified; }

void lz_dp_row16_overflow() { fprintf(stderr, "emul: a 16-bit sweep-row cell out of range\n"); abort(); }
static u64 g_row16_runs = 0;
extern "C" u64 emul_gapped_row16_runs(void) { return g_row16_runs; }
// one DP through lz_dp_run on the sweep-row type SH, the template switches picked at run time
template <class SH> static void emul_run_dp(bool bounded, bool no_trim, bool repl, CpuPhases& x, SH& sh, const LzDpParams& P, const LzDpJob& J, const s32* tab, LzDpResult* r)
{
    if (bounded) { if (no_trim) (repl ? lz_dp_run<true, true, true, CpuPhases, SH> : lz_dp_run<true, true, false, CpuPhases, SH>)(x, sh, P, J, tab, r); else (repl ? lz_dp_run<false, true, true, CpuPhases, SH> : lz_dp_run<false, true, false, CpuPhases, SH>)(x, sh, P, J, tab, r); }
    else         { if (no_trim) (repl ? lz_dp_run<true, false, true, CpuPhases, SH> : lz_dp_run<true, false, false, CpuPhases, SH>)(x, sh, P, J, tab, r); else (repl ? lz_dp_run<false, false, true, CpuPhases, SH> : lz_dp_run<false, false, false, CpuPhases, SH>)(x, sh, P, J, tab, r); }
}

struct EmulExec : LzDpExecutor {
    std::vector<u8> tdp, qdp;           // padded DP-class codes
    u32 tlen, qlen; s32 tab[LZ_NCLASS * LZ_NCLASS];
    s32 gap_e, gap_oe, ydrop; u32 tb_len; s32 no_trim = 0;
    u32 tb_slot;                        // first-try slot size (tests shrink it to exercise the retry)
    u64 retries = 0, wide_runs = 0, piece_reruns = 0;
    int run(const LzHostSnapshot& snap, std::vector<LzDpJob>& jobs, std::vector<LzDpResult>& res,
            std::vector<std::vector<u32>>& ops) override
    {
        static LzDpShared sh; static LzDpShared16 sh16;
        const bool bounded = !snap.aligns.empty();
        // the 16-bit sweep row wherever the product's rule allows it: EMUL_ROW16=1 for every such DP, =0 for none, default: every other pair of DPs
        static const int row16_mode = []() { const char* e = getenv("EMUL_ROW16"); return e ? atoi(e) : -1; }();
        const bool row16_can = lz_dp_row16_ok(ydrop, gap_oe, tab, LZ_NCLASS * LZ_NCLASS);
        static const u32 first_horizon = []() { const char* e = getenv("EMUL_DP_HORIZON"); return (u32)(e ? atoi(e) : 0); }();   // tests: a short one, to run into the re-run
        static const bool verify = getenv("EMUL_VERIFY_PIECES") != nullptr;      // every job's pieces against the reference's row-by-row routines
        for (size_t k = 0; k < jobs.size(); k++) {
            if (verify && bounded)
                for (u32 h : { 61u, 776u, jobs[k].M }) {
                    const u32 bad = emul_verify_pieces(snap, jobs[k], h, std::min<u32>(jobs[k].M, 6000u));
                    if (bad) { fprintf(stderr, "emul: pieces of job %zu (anchor %u %u) differ from the row-by-row routines at row %u (horizon %u)\n", k, jobs[k].anchor1, jobs[k].anchor2, bad, h); return LZGPU_ERR_STATE; }
                    g_pieces_verified++;
                }
            u32 slot = tb_slot;
            u32 horizon = first_horizon ? first_horizon : jobs[k].M;
            for (;;) {
                std::vector<u8> tb(slot); std::vector<u32> rows(slot / 16 + 16), opbuf(slot / 4 + 16);
                LzDpParams P; P.tdp = tdp.data() + LZ_SEQ_PAD; P.tlen = tlen; P.qdp = qdp.data() + LZ_SEQ_PAD; P.qlen = qlen;
                P.gap_e = gap_e; P.gap_oe = gap_oe; P.ydrop = ydrop; P.ydrop_tail = ydrop / gap_e + 6; P.tb_len = tb_len; P.no_trim = no_trim;
                // the job's pieces, as the product's executor lays them out: left bound, right bound, masks
                LzDpPieces pcs; lzh_dp_pieces(snap, jobs[k], horizon, pcs);
                std::vector<LzDpPiece> arena(pcs.lb); arena.insert(arena.end(), pcs.rb.begin(), pcs.rb.end()); arena.insert(arena.end(), pcs.mk.begin(), pcs.mk.end());
                arena.push_back(LzDpPiece{ 0, 0, 0, 0 });
                P.tb_arena = tb.data(); P.row_arena = rows.data(); P.ops_arena = opbuf.data(); P.pc_arena = arena.data();
                LzDpJob& J = jobs[k];
                J.tb_off = 0; J.tb_cap = slot; J.row_off = 0; J.row_cap = (u32)rows.size(); J.ops_off = 0; J.ops_cap = (u32)opbuf.size();
                J.pc_off = 0; J.n_lb = (u32)pcs.lb.size(); J.n_rb = (u32)pcs.rb.size(); J.n_mk = (u32)pcs.mk.size();
                J.horizon = pcs.complete ? 0xFFFFFFFFu : horizon;
                CpuPhases x;
                const bool row16 = row16_can && (row16_mode < 0 ? (k & 2) != 0 : row16_mode != 0);
                if (row16) { emul_run_dp(bounded, no_trim != 0, (k & 1) != 0, x, sh16, P, J, tab, &res[k]); g_row16_runs++; }
                else         emul_run_dp(bounded, no_trim != 0, (k & 1) != 0, x, sh, P, J, tab, &res[k]);
                if (res[k].status == LZ_DP_TOO_WIDE) {              // the product's second kernel: the ring in an HBM slot
                    static std::vector<u8> ring(LzDpRingHbm::SLOT_BYTES);
                    static LzDpSharedWide shw;
                    shw.bind(ring.data());
                    CpuPhases xw;
                    if (bounded) { if (no_trim) lz_dp_run<true, true, false>(xw, shw, P, J, tab, &res[k]); else lz_dp_run<false, true, false>(xw, shw, P, J, tab, &res[k]); }
                    else         { if (no_trim) lz_dp_run<true, false, true>(xw, shw, P, J, tab, &res[k]); else lz_dp_run<false, false, true>(xw, shw, P, J, tab, &res[k]); }
                    wide_runs++;
                }
                if (res[k].status == LZ_DP_PIECE_SLOT) {            // the sweep passed the pieces' horizon: again with more of them
                    horizon = horizon < (1u << 28) ? horizon * 4 : 0xFFFFFFF0u; piece_reruns++;
                    continue;
                }
                if (res[k].status == LZ_DP_TB_SLOT || res[k].status == LZ_DP_ROW_SLOT || res[k].status == LZ_DP_OPS_SLOT) {
                    if (slot >= tb_len) return LZGPU_ERR_STATE;
                    slot = slot * 4 < tb_len ? slot * 4 : tb_len; retries++;
                    continue;
                }
                if (res[k].status != LZ_DP_OK) { fprintf(stderr, "emul: job %zu status %u (row %u LY %u)\n", k, res[k].status, sh.row, sh.LY); return LZGPU_NH_UNSUPPORTED; }
                ops[k].assign(opbuf.begin(), opbuf.begin() + res[k].n_ops);
                break;
            }
        }
        return 0;
    }
};

static void dp_codes(const u8* seq, u32 len, const u8 cls[256], std::vector<u8>& out)
{
    out.assign((size_t)len + 2 * LZ_SEQ_PAD, 0);
    for (u32 i = 0; i < len; i++) out[LZ_SEQ_PAD + i] = cls[seq[i]] & 31;
}

static LzGappedStats g_stats; static u64 g_retries, g_wide, g_piece_reruns;
extern "C" u64 emul_gapped_piece_reruns(void) { return g_piece_reruns; }        // DPs run again because the sweep passed their pieces' horizon (cumulative)
extern "C" void emul_gapped_stats(u64* out) { out[0] = g_stats.anchors; out[1] = g_stats.anchors_extended; out[2] = g_stats.dp_runs;
    out[3] = g_stats.dp_cells; out[4] = g_stats.rounds; out[5] = g_stats.reruns; out[6] = g_retries; out[7] = g_wide; }

static int g_all_bounds = 0, g_no_trim = 0;
extern "C" void emul_gapped_options(int all_bounds, int no_trim) { g_all_bounds = all_bounds; g_no_trim = no_trim; }   // of the calls that follow

extern "C" int emul_gapped_extend(const u8* t, u32 tlen, const u8* q, u32 qlen, const s32* sub,
                                  s32 gap_open, s32 gap_extend, s32 ydrop, s32 score_thresh, u32 tb_len,
                                  lz_segment* anchors, u32 n_anchors, int reduce, u32 window, u32 tb_slot,
                                  lz_align** out, u64* n_out, u32** ops, u64* n_ops)
{
    EmulExec ex;
    ex.no_trim = g_no_trim;
    u8 rowc[256], colc[256];
    int rc = lzh_score_classes(sub, rowc, colc, ex.tab); if (rc) return rc;
    dp_codes(t, tlen, rowc, ex.tdp); dp_codes(q, qlen, colc, ex.qdp);
    ex.tlen = tlen; ex.qlen = qlen; ex.gap_e = gap_extend; ex.gap_oe = gap_open + gap_extend; ex.ydrop = ydrop;
    ex.tb_len = tb_len ? tb_len : 80u * 1024 * 1024; ex.tb_slot = tb_slot ? tb_slot : (1u << 22);
    LzGappedParams G; G.t = t; G.tlen = tlen; G.q = q; G.qlen = qlen; G.sub = sub;
    G.gap_open = gap_open; G.gap_extend = gap_extend; G.ydrop = ydrop; G.score_thresh = score_thresh; G.window = window; G.all_bounds = g_all_bounds != 0;
    if (reduce) lzh_reduce_to_points(t, q, sub, anchors, n_anchors);
    std::vector<lz_align> al; std::vector<u32> op;
    rc = lzh_gapped_extend(G, ex, anchors, n_anchors, al, op, g_stats);
    g_retries = ex.retries; g_wide = ex.wide_runs; g_piece_reruns += ex.piece_reruns;
    if (rc) return rc;
    *out = (lz_align*)malloc((al.size() ? al.size() : 1) * sizeof(lz_align));
    *ops = (u32*)malloc((op.size() ? op.size() : 1) * 4);
    if (!al.empty()) memcpy(*out, al.data(), al.size() * sizeof(lz_align));
    if (!op.empty()) memcpy(*ops, op.data(), op.size() * 4);
    *n_out = al.size(); *n_ops = op.size();
    return 0;
}

extern "C" int emul_selftest_neighbours(u32 seed, u32 n_aligns, u32 n_queries) { return lzh_selftest_neighbours(seed, n_aligns, n_queries); }

