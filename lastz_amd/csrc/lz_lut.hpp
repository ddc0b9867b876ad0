// lz_lut.hpp -- phase A of the seed stage on 2-bit codes: the two X-drop scans of a raw hit
// (xdrop_extend_seed_hit loops 1 and 2, src/seed_search.c:2623-2632, 2684-2693) advance THREE bases per
// step through a look-up table held in LDS, instead of one base per step.
//
// Why this is exact.  Let run / best be the reference's running and best score of one scan and
// m = run - best + xDrop >= 0 its margin (the loop test "run >= best - xDrop" is m >= 0).  For the next
// three bases with prefix sums P1,P2,P3 (P3 = C), minP = min Pj, maxP = max Pj:
//   * the scan stops inside the group  <=>  m + minP < 0, PROVIDED no three-base group can lose more than
//     xDrop from a prefix maximum set inside the same group (checked when the table is built: the "internal
//     drawdown" of every group, i.e. the loss over the one or two bases after a maximum, is <= xDrop; with HOXD70 it is
//     at most 250 against xDrop 910);
//   * otherwise best' = max(best, run + maxP), run' = run + C, i.e. m' = min(m, xDrop - maxP) + C.
// An entry therefore holds A = max(0,-minP), B' = xDrop - max(maxP,0) (m never exceeds xDrop) and C; a step is one LDS read, one compare
// and three integer operations for three bases.  The group in which the scan stops (or which is cut by the
// end of the sequences / a byte that is not A,C,G,T) is walked base by base with the reference's own loop.
//
// Bytes outside the 2-bit alphabet ("specials": lower case, N, the NUL between partitions, ...) are kept in a
// separate 1-bit-per-base mask.  The LUT path is only taken when every special byte that OCCURS in the two
// sequences scores below -xDrop against everything that occurs in the other one: the reference's scan then
// consumes that base and stops without raising its best, which is what lz_lut_window does when it meets a
// mask bit.  (lzh_lut_eligible checks this; any other matrix runs the byte-code scans of lz_common.hpp.)
//
// The functions here are the per-lane device logic (LZ_HD: also compiled for the host by tests/emul).
#pragma once
#include "lz_common.hpp"

#define LZ_PAD2         128          // padding bases in front of base 0 in the 2-bit and mask arrays (and >= that after the end)
#define LZ_LUT_ENTRIES  4096         // 3 bases x (2 + 2) bits
#define LZ_LUT_WIN_G    20           // groups per 16-byte window
#define LZ_LUT_WIN_B    60           // bases per window
#define LZ_LUT_MAXWIN   3            // windows per scan (180 bases); a scan still alive after that makes the hit SLOW

struct LzLutEntry { u32 ab; s32 c; };           // ab = A (u16) | B' (s16) << 16

struct LzLutParams {
    const u8* t2; const u8* q2;                   // 2-bit codes: base i at bits 2*((i+PAD2)&3) of byte (i+PAD2)>>2
    const u8* tsp; const u8* qsp;                 // special masks: base i at bit (i+PAD2)&7 of byte (i+PAD2)>>3 (NULL: no specials)
    s32 xdrop;
};

struct LzLutScan { u32 s; s32 run, best; u32 room, used, alive, nwin; };

#if defined(__HIP_DEVICE_COMPILE__)
#define LZ_WAVE_NONE(x) (__builtin_amdgcn_ballot_w64((bool)(x)) == 0ull)
LZ_HD u32 lz_alignbit(u32 hi, u32 lo, u32 sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
LZ_HD u32 lz_ctz64(u64 x) { return (u32)__builtin_ctzll(x); }
LZ_HD u32 lz_clz64(u64 x) { return (u32)__builtin_clzll(x); }
#else
#define LZ_WAVE_NONE(x) (!(x))
LZ_HD u32 lz_alignbit(u32 hi, u32 lo, u32 sh) { sh &= 31u; return sh ? (lo >> sh) | (hi << (32u - sh)) : lo; }
LZ_HD u32 lz_ctz64(u64 x) { return (u32)__builtin_ctzll(x); }
LZ_HD u32 lz_clz64(u64 x) { return (u32)__builtin_clzll(x); }
#endif

// six bits starting at a compile-time bit position of a 128-bit window (position order: lowest base in the low bits)
LZ_HD u32 lz_lut_field(const u32 w[4], int bit)
{
    const int wd = bit >> 5, off = bit & 31;
    u32 v = w[wd] >> off;
    if (off > 26) v |= w[wd + 1] << (32 - off);
    return v & 63u;
}

// 128-bit window >> sh (sh < 32)
LZ_HD void lz_lut_funnel(const LzVec16& v, u32 sh, u32 w[4])
{
    w[0] = lz_alignbit(v.w[1], v.w[0], sh); w[1] = lz_alignbit(v.w[2], v.w[1], sh);
    w[2] = lz_alignbit(v.w[3], v.w[2], sh); w[3] = v.w[3] >> sh;
}

// how many bases from position s onwards (RIGHT) / from s-1 downwards (!RIGHT) are plain in this sequence: 0..64
template <bool RIGHT>
LZ_HD u64 lz_lut_mask64(const u8* sp, s64 s)
{
    if (RIGHT) {
        const u64 b = (u64)(s + LZ_PAD2);
        const LzVec16 v = lz_load16(sp + (b >> 3));
        const u32 k = (u32)(b & 7u);
        const u32 lo = lz_alignbit(v.w[1], v.w[0], k), hi = lz_alignbit(v.w[2], v.w[1], k);
        return ((u64)hi << 32) | lo;                    // bit j = base s + j
    }
    const u64 b = (u64)(s - 1 + LZ_PAD2);
    const LzVec16 v = lz_load16(sp + (b >> 3) - 14);   // base s-1 is bit 112 + (b & 7) of the 128 loaded bits
    const u32 k = 17u + (u32)(b & 7u);                 // 17..24: bit 112+k' of v = bit 80+k' of (v >> 32) -> bit 63
    const u32 lo = lz_alignbit(v.w[2], v.w[1], k), hi = lz_alignbit(v.w[3], v.w[2], k);
    return ((u64)hi << 32) | lo;                        // bit 63 - j = base s - 1 - j
}

#define LZ_LUT_STEP(K, IDX, E)                                                                        \
    if (go) {                                                                                        \
        if (3u * ((K) + 1u) > lim) { go = false; gx = (K); ix = (IDX); }                             \
        else if ((u32)m < ((E).ab & 0xFFFFu)) { go = false; fail = true; gx = (K); ix = (IDX); }     \
        else { const s32 b_ = (s32)(E).ab >> 16; m = (m < b_ ? m : b_) + (E).c; run += (E).c; }    \
    }

// One 16-byte window (up to 60 bases) of one scan.  lut = this direction's table, m16 = the 4 x 4 matrix.
// On return st.alive says whether the scan goes on into the next window.
template <bool RIGHT, bool SPECIAL>
LZ_HD void lz_lut_window(const LzLutParams& P, const LzLutEntry* lut, const s32* m16, s32 diag, LzLutScan& st)
{
    const s32 X = P.xdrop;
    const s64 s = (s64)st.s, sq = s - (s64)diag;
    u32 tw[4], qw[4];
    if (RIGHT) {
        const u64 bt = (u64)(s + LZ_PAD2), bq = (u64)(sq + LZ_PAD2);
        const LzVec16 tv = lz_load16(P.t2 + (bt >> 2)), qv = lz_load16(P.q2 + (bq >> 2));
        lz_lut_funnel(tv, 2u * (u32)(bt & 3u), tw);    // base s at bit 0
        lz_lut_funnel(qv, 2u * (u32)(bq & 3u), qw);
    } else {
        const u64 bt = (u64)(s - 1 + LZ_PAD2), bq = (u64)(sq - 1 + LZ_PAD2);
        const LzVec16 tv = lz_load16(P.t2 + (bt >> 2) - 15), qv = lz_load16(P.q2 + (bq >> 2) - 15);
        lz_lut_funnel(tv, 2u * (u32)(bt & 3u), tw);    // base s-1 at bits 120-121
        lz_lut_funnel(qv, 2u * (u32)(bq & 3u), qw);
    }
    u32 lim = st.room < (u32)LZ_LUT_WIN_B ? st.room : (u32)LZ_LUT_WIN_B;
    bool soft = false;
    if (SPECIAL) {
        const u64 sm = lz_lut_mask64<RIGHT>(P.tsp, s) | lz_lut_mask64<RIGHT>(P.qsp, sq);
        const u32 nsp = sm ? (RIGHT ? lz_ctz64(sm) : lz_clz64(sm)) : 64u;
        if (nsp < lim) { soft = true; lim = nsp; }
    }
    s32 run = st.run, m = run - st.best + X;
    bool go = true, fail = false;
    u32 gx = LZ_LUT_WIN_G, ix = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int blk = 0; blk < LZ_LUT_WIN_G / 4; blk++) {
        if (LZ_WAVE_NONE(go)) break;
        if (go) {
            u32 idx[4]; LzLutEntry e[4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int k = 0; k < 4; k++) {
                const int g = 4 * blk + k;
                const int bit = RIGHT ? 6 * g : 116 - 6 * g;
                idx[k] = (lz_lut_field(tw, bit) << 6) | lz_lut_field(qw, bit);
                e[k] = lut[idx[k]];
            }
            LZ_LUT_STEP(4u * blk + 0u, idx[0], e[0])
            LZ_LUT_STEP(4u * blk + 1u, idx[1], e[1])
            LZ_LUT_STEP(4u * blk + 2u, idx[2], e[2])
            LZ_LUT_STEP(4u * blk + 3u, idx[3], e[3])
        }
    }
    // the group the fast loop stopped in, base by base (the reference's loop)
    u32 r = 0;
    if (gx < (u32)LZ_LUT_WIN_G) r = fail ? 3u : lim - 3u * gx;         // (lim - 3 gx is 0..2 when the limit cut the group)
    s32 best = run - m + X;
    bool dead = false; u32 j = 0;
    const u32 tf = ix >> 6, qf = ix & 63u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (u32 b = 0; b < 3; b++) {
        if (!dead && b < r) {
            const u32 sh = RIGHT ? 2u * b : 2u * (2u - b);
            run += m16[(((tf >> sh) & 3u) << 2) | ((qf >> sh) & 3u)];
            j++;
            if (run > best) best = run;
            if (run < best - X) dead = true;
        }
    }
    st.run = run; st.best = best;
    st.used += 3u * gx + j;
    st.nwin++;
    if (dead) st.alive = 0;
    else if (fail) st.alive = 2;                                        // cannot happen with an eligible table; 2 = "undecided", the hit becomes SLOW
    else if (soft) { st.used += 1u; st.alive = 0; }                     // the special base is consumed and ends the scan
    else if (lim == st.room) st.alive = 0;                              // end of a sequence / the left stop
    else { st.alive = 1; st.room -= (u32)LZ_LUT_WIN_B; st.s = RIGHT ? st.s + (u32)LZ_LUT_WIN_B : st.s - (u32)LZ_LUT_WIN_B; }
}

// scan set-up of one raw hit (diagEnd == 0, as in lz_probe_head)
LZ_HD void lz_lut_init(u64 key, u32 tlen, u32 qlen, s32& diag, LzLutScan& L, LzLutScan& R)
{
    const u32 pos2 = (u32)key;
    diag = (s32)(u32)(key >> 32);
    const u32 pos1 = pos2 + (u32)diag;
    const s32 stopl = diag > 0 ? diag : 0;
    const s32 stopr = ((s32)tlen <= (s32)qlen + diag) ? (s32)tlen : (s32)qlen + diag;
    L.s = R.s = pos1; L.run = L.best = R.run = R.best = 0; L.used = R.used = 0; L.nwin = R.nwin = 0;
    L.room = (u32)((s32)pos1 - stopl); R.room = (u32)(stopr - (s32)pos1);
    L.alive = ((s32)pos1 > stopl) ? 1u : 0u; R.alive = ((s32)pos1 < stopr) ? 1u : 0u;
}

// the 4-byte phase-A summary (same meaning as lz_probe_summary)
LZ_HD u32 lz_lut_summary(const LzLutScan& L, const LzLutScan& R, s32 min_score)
{
    u32 summ = (L.used & 0xFFu) | ((R.used & 0xFFu) << 8);
    if (L.alive || R.alive || L.best + R.best >= min_score) summ |= LZ_SUMM_SLOW;
    return summ;
}

template <bool SPECIAL>
LZ_HD u32 lz_lut_probe_hit(const LzLutParams& P, const LzLutEntry* lut_r, const LzLutEntry* lut_l, const s32* m16,
                           u32 tlen, u32 qlen, s32 min_score, u64 key)
{
    s32 diag; LzLutScan L, R;
    lz_lut_init(key, tlen, qlen, diag, L, R);
    while (L.alive == 1 && L.nwin < (u32)LZ_LUT_MAXWIN) lz_lut_window<false, SPECIAL>(P, lut_l, m16, diag, L);
    while (R.alive == 1 && R.nwin < (u32)LZ_LUT_MAXWIN) lz_lut_window<true, SPECIAL>(P, lut_r, m16, diag, R);
    return lz_lut_summary(L, R, min_score);
}

// ---- the hit record that travels from phase A to phase B (one u64 per raw hit, partitioned by the high
// 8 bits of hashedDiag, in discovery order inside a partition):
//   bits  0..30  pos2 (end of the seed word in the query)
//   bits 31..38  low 8 bits of hashedDiag (the bucket inside its partition)
//   bits 39..54  fast hits: bases the left scan consumed | bases the right scan consumed << 8
//                SLOW hits: bits 16..31 of the diagonal pos1 - pos2 (the low 16 are the bucket)
//   bit  63      SLOW
LZ_HD u64 lz_hit_record(u64 key, u32 summ)
{
    const u32 pos2 = (u32)key, diag = (u32)(key >> 32);
    const bool slow = (summ & LZ_SUMM_SLOW) != 0;
    const u64 payload = slow ? (u64)(diag >> 16) : (u64)(summ & 0xFFFFu);
    return (u64)pos2 | ((u64)(diag & 0xFFu) << 31) | (payload << 39) | (slow ? (1ull << 63) : 0ull);
}
#define LZ_REC_POS2(r)    ((u32)((r) & 0x7FFFFFFFu))
#define LZ_REC_LOW8(r)    ((u32)((r) >> 31) & 0xFFu)
#define LZ_REC_PAYLOAD(r) ((u32)((r) >> 39) & 0xFFFFu)
#define LZ_REC_SLOW(r)    ((u32)((r) >> 63))

// ---- phase B: one record of a bucket's stream, in discovery order, with diagEnd[h] in `dend`.
// This is process_for_simple_hit + xdrop_extend_seed_hit (src/seed_search.c:1056-1192, 2528-2959) for hit
// (pos1 = pos2 + diag, pos2): the diagEnd test (:1113), then either the phase-A summary (the unclipped scans
// scored below the threshold: a left scan clipped at diagEnd walks a prefix of the same bases -- its best can
// only be lower, still no HSP -- and the right scan does not depend on it, so only the count of bases differs,
// min(unclipped, room)) or, for SLOW records, the real extension.
template <class Emit>
LZ_HD void lz_settle_record(const LzExtendParams& P, const s32* score_tab, u64 rec, u32 h /*hashedDiag of the bucket*/,
                            u32& dend, u64& n_ext, u64& n_bp, Emit&& emit)
{
    const u32 p2 = LZ_REC_POS2(rec);
    if (dend > p2 - P.seed_len) return;                         // :1113
    n_ext++;
    const u32 pay = LZ_REC_PAYLOAD(rec);
    if (!LZ_REC_SLOW(rec)) {
        const u32 room = p2 - dend, dlo = pay & 0xFFu, dext = pay >> 8;
        const u32 extent = p2 + dext;                           // :2785
        n_bp += (dlo < room ? dlo : room) + dext;               // :2818
        if (extent > dend) dend = extent;
        return;
    }
    const s32 diag = (s32)((pay << 16) | h);
    dend = lz_reextend(P, score_tab, p2, diag, dend, n_bp, emit);
}
