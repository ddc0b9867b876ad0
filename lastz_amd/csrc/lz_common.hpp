// lz_common.hpp -- shared POD types and the per-thread ("lane") logic of the seed stage.
//
// The functions marked LZ_HD are what one GPU thread executes; the __global__ kernels in
// seed_kernels.hip are thin wrappers that map threadIdx/blockIdx onto them.  They are written
// against plain pointers so that tests/emul/ can also run them on the host (one "lane" at a
// time) to check the decomposition logic without a GPU.  That host harness is test
// infrastructure; the product library never executes these functions on the CPU.
#pragma once
#include <stdint.h>

#if defined(__HIP__)            // clang in HIP mode (device + host passes of a .hip file)
#define LZ_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define LZ_HD inline
#endif

typedef uint8_t  u8;
typedef uint32_t u32;
typedef int32_t  s32;
typedef uint64_t u64;
typedef int64_t  s64;

#define LZ_MAX_PARTS   16
#define LZ_MAX_PROBES  128
#define LZ_DIAG_BITS   16                       // diagHashSize = 65536, src/diag_hash.h:56
#define LZ_DIAG_SIZE   (1u << LZ_DIAG_BITS)
#define LZ_NCLASS      32                       // score classes per axis (5 bits of the code byte)
#define LZ_SEQ_PAD     64                       // zero bytes before and after every device sequence

// One byte per base on the device:
//   bits 0-4  score class (row class for the target, column class for the query)
//   bits 5-6  2-bit nucleotide from charToBits (src/dna_utilities.c:56-94)
//   bit  7    set when charToBits[b] < 0 (byte cannot be part of a seed word)
#define LZ_CODE_CLASS(c)   ((c) & 31u)
#define LZ_CODE_BITS(c)    (((c) >> 5) & 3u)
#define LZ_CODE_INVALID    0x80u

struct LzSeedDev {
    s32 length, weight, nparts;
    s32 shift[LZ_MAX_PARTS];
    u32 mask[LZ_MAX_PARTS];
    s32 nprobes;
    u32 probe_xor[LZ_MAX_PROBES];
};

// candidate HSP produced by the extension kernel (host finishes entropy + ordering)
struct LzHspRec {
    u32 seed_pos1, seed_pos2;   // the raw seed hit (end positions)
    u32 end1;                   // rightStop in the target (exclusive)
    u32 length;                 // rightStop - leftStart
    s32 score;                  // leftScore + rightScore, before the entropy adjustment
};

// apply_seed, src/seeds.c:1373-1376
LZ_HD u32 lz_apply_seed(const LzSeedDev& sd, u64 w)
{
    u32 packed = 0;
    for (int p = 0; p < sd.nparts; p++) packed |= (u32)(w >> sd.shift[p]) & sd.mask[p];
    return packed;
}

// The seed word whose window is code[pos-L .. pos).  Returns false if any byte of the window
// cannot be in a word (the reference restarts its rolling window there,
// src/seed_search.c:499-510, src/pos_table.c:436-447).
LZ_HD bool lz_window_word(const u8* code, u32 pos, const LzSeedDev& sd, u32& packed)
{
    u64 w = 0; u32 bad = 0;
    const u8* p = code + pos - (u32)sd.length;
    for (int k = 0; k < sd.length; k++) { u32 c = p[k]; bad |= c; w = (w << 2) | LZ_CODE_BITS(c); }
    if (bad & LZ_CODE_INVALID) return false;
    packed = lz_apply_seed(sd, w);
    return true;
}

// Number of raw seed hits the query word ending at pos2 generates: the sum over probes of the
// CSR list length (find_table_matches, src/seed_search.c:823-832).
LZ_HD u32 lz_count_hits_at(const u8* qcode, u32 pos2, u32 lo, const LzSeedDev& sd, const u32* wstart, bool& valid, u32& packed)
{
    valid = false;
    if (pos2 < lo + (u32)sd.length) return 0;            // window must start at or after the interval start
    if (!lz_window_word(qcode, pos2, sd, packed)) return 0;
    valid = true;
    u32 n = 0;
    for (int p = 0; p < sd.nprobes; p++) { u32 w = packed ^ sd.probe_xor[p]; n += wstart[w + 1] - wstart[w]; }
    return n;
}

// hit key: high word = diagonal (pos1-pos2, two's complement), low word = pos2.  Bits 32..47 are
// hashedDiag(pos1,pos2) (src/diag_hash.h:61-62), the only bits the bucket sort looks at.
LZ_HD u64 lz_hit_key(u32 pos1, u32 pos2) { return ((u64)(u32)(pos1 - pos2) << 32) | (u64)pos2; }

// Write the hits of the query word ending at pos2, in the reference's enumeration order
// (probe order, then chain order = descending pos1), to out[0..n).
LZ_HD void lz_fill_hits_at(const u8* qcode, u32 pos2, const LzSeedDev& sd, const u32* wstart, const u32* wpos, u64* out)
{
    u32 packed;
    if (!lz_window_word(qcode, pos2, sd, packed)) return;
    for (int p = 0; p < sd.nprobes; p++) {
        u32 w = packed ^ sd.probe_xor[p];
        u32 a = wstart[w], b = wstart[w + 1];
        for (u32 j = a; j < b; j++) *out++ = lz_hit_key(wpos[j], pos2);
    }
}

struct LzExtendParams {
    const u8* tcode; u32 tlen;     // target codes (tcode[0] is base 0; LZ_SEQ_PAD readable bytes either side)
    const u8* qcode; u32 qlen;
    s32 xdrop;
    s32 min_score;                 // candidates with left+right >= min_score are emitted
    u32 seed_len;
    u32 cls8;                      // every scoring class of either sequence is < 8: lz_scan16_fast is usable
    const u8* tnib; const u8* qnib; // 4-bit class codes, base i = nibble i + LZ_SEQ_PAD (NULL unless cls8)
};

struct LzVec16 { u32 w[4]; };
// 16 consecutive code bytes from an arbitrary (unaligned) address: one global_load_dwordx4 on
// gfx950 (amdhsa enables unaligned access mode); the sequences carry LZ_SEQ_PAD readable bytes on
// both sides so a block may overhang either end.
LZ_HD LzVec16 lz_load16(const u8* p) { LzVec16 v; __builtin_memcpy(&v, p, 16); return v; }
#define LZ_VBYTE(v, k) (((v).w[(k) >> 2] >> (((k) & 3) * 8)) & 0xFFu)

// ---- phase A: every raw hit, independently (no diagEnd), scanned at most LZ_PROBE_CAP bases per
// side.  The reference's two X-drop loops depend on the diagonal hash only through the left stop
// position max(0, diagEnd[h]+diag) (:2612-2616); a scan that terminates on its own at query
// position `lo` is therefore the scan the reference performs whenever diagEnd[h] <= lo.  The
// summary lets the serial per-bucket pass (phase B) settle such hits without touching the
// sequences; anything else (still alive at the cap, or scoring >= min_score so that an HSP
// record is needed) is flagged SLOW and re-done in order by phase B.
#define LZ_PROBE_CAP   128            // multiple of 16
#define LZ_SUMM_SLOW   0x10000u
#define LZ_FAST_GROUP  16             // (key, summary) pairs loaded together: phase B runs one wave per SIMD, so
                                     // the bytes in flight per lane are what hides the memory latency
#define LZ_FAST_RUN    4              // groups settled per trip of the phase-B loop
#define LZ_SUMM_DLO(s)  ((s) & 0xFFu)         // pos2 - lo   (bases the left scan consumed)
#define LZ_SUMM_DEXT(s) (((s) >> 8) & 0xFFu)  // extent - pos2 (bases the right scan consumed)

// A whole block in one go, for the common case (>= 16 bases of room, < 8 scoring classes).
// (tab8 is always the table's address and `fast` the switch, so that on the device the LDS address of the
// table is a constant the look-ups fold into their offset field: one VALU add less per base.)
// - score look-up: (row class << 5 | column class << 2) IS the byte address in an 8 x 8 table,
//   built four bases at a time; with 8 words per row the 16 ACGT pairs fall into 16 LDS banks.
// - X-drop chain without masks: a base passes if run >= best - xDrop (the test of the reference
//   with best not yet updated -- the same thing, as xDrop >= 0); the first base that fails is
//   replaced by a poison value that makes every later base fail too and never raises best.
// Returns the number of bases that passed (16: the scan goes on; otherwise the scan consumed one
// more base, the failing one, and stopped).  run is meaningless after a failure.
#define LZ_SCAN_POISON (-(1 << 30))
LZ_HD u32 lz_xdrop_chain16(const s32 sc[16], s32 xd, s32& run, s32& best)
{
    s32 p = run, thr = best - xd;
    u32 nok = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 16; k++) {
        p += sc[k];
        const bool ok = p >= thr;
        nok += ok ? 1u : 0u;
        const s32 t = p - xd;
        thr = t > thr ? t : thr;
        p = ok ? p : LZ_SCAN_POISON;
    }
    run = p; best = thr + xd;
    return nok;
}

template <bool REV>
LZ_HD u32 lz_scan16_fast(const s32* tab8, s32 xd, const LzVec16& tv, const LzVec16& qv, s32& run, s32& best)
{
    u32 x[4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 4; j++) x[j] = ((tv.w[j] & 0x07070707u) << 5) | ((qv.w[j] & 0x07070707u) << 2);
    s32 sc[16];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 16; k++) {
        const int b = REV ? 15 - k : k;
        sc[k] = *(const s32*)((const u8*)tab8 + ((x[b >> 2] >> ((b & 3) * 8)) & 0xFFu));
    }
    return lz_xdrop_chain16(sc, xd, run, best);
}

// The same block from 4-bit class codes, two bases per byte (tn[0], tn[1] = 16 target nibbles in
// sequence order, lowest nibble first; likewise qn): phase A is bound by its 16-byte gathers as much
// as by its arithmetic, and packed codes halve the gathers.  Even and odd nibbles are turned into
// table addresses separately (four bases per instruction pair again).
template <bool REV>
LZ_HD u32 lz_scan16_nib(const s32* tab8, s32 xd, const u32 tn[2], const u32 qn[2], s32& run, s32& best)
{
    u32 xe[2], xo[2];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int d = 0; d < 2; d++) {
        xe[d] = ((tn[d] & 0x07070707u) << 5) | ((qn[d] & 0x07070707u) << 2);
        xo[d] = ((tn[d] << 1) & 0xE0E0E0E0u) | ((qn[d] >> 2) & 0x1C1C1C1Cu);
    }
    s32 sc[16];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 16; k++) {
        const int m = REV ? 15 - k : k;                         // nibble of the block
        const u32 w = (m & 1) ? xo[m >> 3] : xe[m >> 3];
        sc[k] = *(const s32*)((const u8*)tab8 + ((w >> (((m & 7) >> 1) * 8)) & 0xFFu));
    }
    return lz_xdrop_chain16(sc, xd, run, best);
}

// One 16-base block of the left scan (loop 1, :2623-2632: bases sl-1, sl-2, ... taken from the
// 16 bytes that END at sl) and of the right scan (loop 2, :2684-2693: the 16 bytes that START at sr).
// "run >= best - xDrop" gates each further base; the return value says whether the scan goes on.
LZ_HD bool lz_scan_left16(const s32* score_tab, const s32* tab8, bool fast, s32 xd, const LzVec16& tv, const LzVec16& qv,
                          s32 stopl, u32& sl, s32& runl, s32& bestl)
{
    const u32 room = (u32)((s32)sl - stopl);
    if (fast && room >= 16u && xd >= 0) {
        const u32 nok = lz_scan16_fast<true>(tab8, xd, tv, qv, runl, bestl);
        sl -= (nok < 16u) ? nok + 1u : 16u;
        return nok == 16u && (s32)sl > stopl;
    }
    s32 sc[16];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 16; k++)
        sc[k] = score_tab[(LZ_CODE_CLASS(LZ_VBYTE(tv, 15 - k)) << 5) | LZ_CODE_CLASS(LZ_VBYTE(qv, 15 - k))];
    bool go = true;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 16; k++)
        if (go && (u32)k < room) { runl += sc[k]; --sl; if (runl > bestl) bestl = runl; go = runl >= bestl - xd; }
    return go && ((s32)sl > stopl);
}
LZ_HD bool lz_scan_right16(const s32* score_tab, const s32* tab8, bool fast, s32 xd, const LzVec16& tv, const LzVec16& qv,
                           s32 stopr, u32& sr, s32& runr, s32& bestr)
{
    const u32 room = (u32)(stopr - (s32)sr);
    if (fast && room >= 16u && xd >= 0) {
        const u32 nok = lz_scan16_fast<false>(tab8, xd, tv, qv, runr, bestr);
        sr += (nok < 16u) ? nok + 1u : 16u;
        return nok == 16u && (s32)sr < stopr;
    }
    s32 sc[16];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 16; k++)
        sc[k] = score_tab[(LZ_CODE_CLASS(LZ_VBYTE(tv, k)) << 5) | LZ_CODE_CLASS(LZ_VBYTE(qv, k))];
    bool go = true;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 16; k++)
        if (go && (u32)k < room) { runr += sc[k]; ++sr; if (runr > bestr) bestr = runr; go = runr >= bestr - xd; }
    return go && ((s32)sr < stopr);
}

// Phase A of one hit, in two parts.
// lz_probe_head: the address of every block is known from the key alone, so the first
// LZ_PROBE_AHEAD_L / _R blocks of both sequences are loaded before any of them is scored (two
// dependent memory round trips per hit: key, blocks) and scanned; most hits end inside them.
// lz_scan_continue: a scan that is still going continues block by block, up to the cap.  The two
// scans of a hit are independent (loop 2 restarts at the seed end, :2663-2682), so the kernel hands
// the unfinished ones to other lanes as separate tasks (seed_kernels.hip); lz_probe_hit is the plain
// composition.  (The 64 bytes of padding around the sequences cover blocks that reach over an end.)
#ifndef LZ_PROBE_AHEAD_L
#define LZ_PROBE_AHEAD_L 3
#define LZ_PROBE_AHEAD_R 2
#endif
// 4-bit codes: nibble n of the array is base n - LZ_SEQ_PAD.  `cnt` dwords (8 bases each) starting AT
// base `base` (any parity): whole 16-byte loads from the byte that holds it, then a 4-bit funnel shift.
// (LZ_PIN keeps the compiler from splitting the 16-byte loads into per-use dword loads sunk into the
// blocks that need them: the point is to have the whole window in flight at once)
#if defined(__HIP_DEVICE_COMPILE__)
#define LZ_PIN(x) asm volatile("" : "+v"(x))
#else
#define LZ_PIN(x) ((void)0)
#endif
template <int NLOAD>
LZ_HD u32 lz_nib_issue(const u8* nib, s64 base, u32* raw /*[4*NLOAD]*/)      // the loads; returns the funnel shift
{
    const s64 n = base + LZ_SEQ_PAD;
    const u8* p = nib + (n >> 1);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < NLOAD; k++) { const LzVec16 v = lz_load16(p + 16 * k); raw[4 * k] = v.w[0]; raw[4 * k + 1] = v.w[1]; raw[4 * k + 2] = v.w[2]; raw[4 * k + 3] = v.w[3]; }
    return (u32)(n & 1) * 4u;
}
template <int NLOAD>
LZ_HD void lz_nib_finish(u32* raw /*[4*NLOAD]*/, u32 sh, u32* out /*[4*NLOAD - 1]*/)
{
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 4 * NLOAD; j++) LZ_PIN(raw[j]);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 4 * NLOAD - 1; j++) out[j] = (u32)(((((u64)raw[j + 1]) << 32) | raw[j]) >> sh);
}
template <int NLOAD>
LZ_HD void lz_load_nib(const u8* nib, s64 base, u32* out /*[4*NLOAD - 1]*/)
{
    u32 raw[4 * NLOAD];
    const u32 sh = lz_nib_issue<NLOAD>(nib, base, raw);
    lz_nib_finish<NLOAD>(raw, sh, out);
}

#define LZ_PROBE_NLOAD ((16 * (LZ_PROBE_AHEAD_L + LZ_PROBE_AHEAD_R) + 1 + 31) / 32)   // 16-byte loads covering the window at either parity
struct LzProbeSt { u32 pos1; s32 diag, stopl, stopr; u32 sl, sr; s32 runl, bestl, runr, bestr; bool alive_l, alive_r; };

LZ_HD void lz_probe_head(const LzExtendParams& P, const s32* score_tab, const s32* tab8 /*8x8, used if fast*/, bool fast, u64 key, LzProbeSt& st)
{
    const s32 xd = P.xdrop;
    const u32 pos2 = (u32)key;
    const s32 diag = (s32)(u32)(key >> 32);
    const u32 pos1 = pos2 + (u32)diag;
    st.pos1 = pos1; st.diag = diag;
    st.stopl = diag > 0 ? diag : 0;                                                      // diagEnd == 0
    st.stopr = ((s32)P.tlen <= (s32)P.qlen + diag) ? (s32)P.tlen : (s32)P.qlen + diag;
    st.sl = st.sr = pos1;
    st.runl = st.bestl = st.runr = st.bestr = 0;
    st.alive_l = ((s32)st.sl > st.stopl) && (0 >= -xd);
    st.alive_r = ((s32)st.sr < st.stopr) && (0 >= -xd);
    if (fast && P.tnib && xd >= 0 && (s32)pos1 - st.stopl >= 16 * LZ_PROBE_AHEAD_L && st.stopr - (s32)pos1 >= 16 * LZ_PROBE_AHEAD_R) {
        // every block of the window is a whole one: LZ_PROBE_NLOAD 16-byte loads per sequence
        u32 tw[4 * LZ_PROBE_NLOAD - 1], qw[4 * LZ_PROBE_NLOAD - 1], traw[4 * LZ_PROBE_NLOAD], qraw[4 * LZ_PROBE_NLOAD];
        const u32 tsh = lz_nib_issue<LZ_PROBE_NLOAD>(P.tnib, (s64)pos1 - 16 * LZ_PROBE_AHEAD_L, traw);
        const u32 qsh = lz_nib_issue<LZ_PROBE_NLOAD>(P.qnib, (s64)pos2 - 16 * LZ_PROBE_AHEAD_L, qraw);
        lz_nib_finish<LZ_PROBE_NLOAD>(traw, tsh, tw);
        lz_nib_finish<LZ_PROBE_NLOAD>(qraw, qsh, qw);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int b = 0; b < LZ_PROBE_AHEAD_L; b++) {
            if (st.alive_l) {
                const int d = 2 * (LZ_PROBE_AHEAD_L - 1 - b);
                const u32 nok = lz_scan16_nib<true>(tab8, xd, tw + d, qw + d, st.runl, st.bestl);
                st.sl -= (nok < 16u) ? nok + 1u : 16u;
                st.alive_l = nok == 16u && (s32)st.sl > st.stopl;
            }
            if (b < LZ_PROBE_AHEAD_R && st.alive_r) {
                const int d = 2 * (LZ_PROBE_AHEAD_L + b);
                const u32 nok = lz_scan16_nib<false>(tab8, xd, tw + d, qw + d, st.runr, st.bestr);
                st.sr += (nok < 16u) ? nok + 1u : 16u;
                st.alive_r = nok == 16u && (s32)st.sr < st.stopr;
            }
        }
        return;
    }
    const u8* tp = P.tcode + pos1;
    const u8* qp = P.qcode + pos2;
    LzVec16 tl[LZ_PROBE_AHEAD_L], ql[LZ_PROBE_AHEAD_L], tr[LZ_PROBE_AHEAD_R], qr[LZ_PROBE_AHEAD_R];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int b = 0; b < LZ_PROBE_AHEAD_L; b++) { tl[b] = lz_load16(tp - 16 * (b + 1)); ql[b] = lz_load16(qp - 16 * (b + 1)); }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int b = 0; b < LZ_PROBE_AHEAD_R; b++) { tr[b] = lz_load16(tp + 16 * b); qr[b] = lz_load16(qp + 16 * b); }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int b = 0; b < LZ_PROBE_AHEAD_L; b++) {
        if (st.alive_l) st.alive_l = lz_scan_left16(score_tab, tab8, fast, xd, tl[b], ql[b], st.stopl, st.sl, st.runl, st.bestl);
        if (b < LZ_PROBE_AHEAD_R && st.alive_r) st.alive_r = lz_scan_right16(score_tab, tab8, fast, xd, tr[b], qr[b], st.stopr, st.sr, st.runr, st.bestr);
    }
}

// continue one scan for at most `blocks` further blocks; returns whether it is still alive (at the cap)
template <bool RIGHT>
LZ_HD bool lz_scan_continue(const LzExtendParams& P, const s32* score_tab, const s32* tab8, bool fast, s32 diag, s32 stop,
                            u32& s, s32& run, s32& best, int blocks)
{
    bool alive = true;
    for (; blocks > 0 && alive; blocks--) {
        const u32 room = RIGHT ? (u32)(stop - (s32)s) : (u32)((s32)s - stop);
        if (fast && P.tnib && room >= 16u && P.xdrop >= 0) {
            u32 tn[3], qn[3];
            const s64 b1 = RIGHT ? (s64)s : (s64)s - 16;
            u32 traw[4], qraw[4];
            const u32 tsh = lz_nib_issue<1>(P.tnib, b1, traw), qsh = lz_nib_issue<1>(P.qnib, b1 - diag, qraw);
            lz_nib_finish<1>(traw, tsh, tn);
            lz_nib_finish<1>(qraw, qsh, qn);
            const u32 nok = lz_scan16_nib<!RIGHT>(tab8, P.xdrop, tn, qn, run, best);
            const u32 c = nok < 16u ? nok + 1u : 16u;
            s = RIGHT ? s + c : s - c;
            alive = nok == 16u && (RIGHT ? (s32)s < stop : (s32)s > stop);
            continue;
        }
        if (RIGHT) alive = lz_scan_right16(score_tab, tab8, fast, P.xdrop, lz_load16(P.tcode + s), lz_load16(P.qcode + ((s32)s - diag)), stop, s, run, best);
        else       alive = lz_scan_left16(score_tab, tab8, fast, P.xdrop, lz_load16(P.tcode + s - 16), lz_load16(P.qcode + ((s32)s - diag) - 16), stop, s, run, best);
    }
    return alive;
}

LZ_HD u32 lz_probe_summary(const LzExtendParams& P, const LzProbeSt& st)
{
    u32 summ = (st.pos1 - st.sl) | ((st.sr - st.pos1) << 8);
    if (st.alive_l || st.alive_r || st.bestl + st.bestr >= P.min_score) summ |= LZ_SUMM_SLOW;
    return summ;
}

LZ_HD u32 lz_probe_hit(const LzExtendParams& P, const s32* score_tab, const s32* tab8 /*8x8, used if fast*/, bool fast, u64 key)
{
    LzProbeSt st;
    lz_probe_head(P, score_tab, tab8, fast, key, st);
    if (st.alive_l) st.alive_l = lz_scan_continue<false>(P, score_tab, tab8, fast, st.diag, st.stopl, st.sl, st.runl, st.bestl, LZ_PROBE_CAP / 16 - LZ_PROBE_AHEAD_L);
    if (st.alive_r) st.alive_r = lz_scan_continue<true>(P, score_tab, tab8, fast, st.diag, st.stopr, st.sr, st.runr, st.bestr, LZ_PROBE_CAP / 16 - LZ_PROBE_AHEAD_R);
    return lz_probe_summary(P, st);
}

// ---- phase B, the slow path.  One raw hit whose phase-A summary does not settle it (an HSP candidate, or a
// scan that ran into the phase-A cap): loops 1 and 2 of xdrop_extend_seed_hit with the bucket's real diagEnd
// (src/seed_search.c:2612-2616, 2623-2632, 2684-2693), 16 bases per trip on each side (the two scans are
// independent of each other: loop 2 restarts from the seed end with runScore = 0, :2663-2682).  Returns the
// bucket's new diagEnd (:2785-2789); emit(rec) is called when leftScore + rightScore >= min_score.
template <class Emit>
LZ_HD u32 lz_reextend(const LzExtendParams& P, const s32* score_tab /*[32*32]*/, u32 pos2, s32 diag, u32 dend,
                      u64& n_bp, Emit&& emit)
{
    const s32 xd = P.xdrop;
    const u32 pos1 = pos2 + (u32)diag;
    s32 stopl = (s32)dend + diag;  if (stopl < 0) stopl = 0;                                 // :2612-2616
    const s32 stopr = ((s32)P.tlen <= (s32)P.qlen + diag) ? (s32)P.tlen : (s32)P.qlen + diag; // :2675-2677
    u32 sl = pos1, sr = pos1, left_start = pos1, right_stop = pos1;
    s32 runl = 0, bestl = 0, runr = 0, bestr = 0;
    bool alive_l = ((s32)sl > stopl) && (0 >= -xd);
    bool alive_r = ((s32)sr < stopr) && (0 >= -xd);
    while (alive_l || alive_r) {
        if (alive_l) {                                          // loop 1, :2623-2632, 16 bases
            const u32 room = (u32)((s32)sl - stopl);
            const LzVec16 tv = lz_load16(P.tcode + sl - 16);
            const LzVec16 qv = lz_load16(P.qcode + ((s32)sl - diag) - 16);
            s32 sc[16];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int k = 0; k < 16; k++)
                sc[k] = score_tab[(LZ_CODE_CLASS(LZ_VBYTE(tv, 15 - k)) << 5) | LZ_CODE_CLASS(LZ_VBYTE(qv, 15 - k))];
            bool go = true;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int k = 0; k < 16; k++) {
                if (go && (u32)k < room) {
                    runl += sc[k];
                    --sl;
                    if (runl > bestl) { bestl = runl; left_start = sl; }
                    go = runl >= bestl - xd;
                }
            }
            alive_l = go && ((s32)sl > stopl);
        }
        if (alive_r) {                                          // loop 2, :2684-2693, 16 bases
            const u32 room = (u32)(stopr - (s32)sr);
            const LzVec16 tv = lz_load16(P.tcode + sr);
            const LzVec16 qv = lz_load16(P.qcode + ((s32)sr - diag));
            s32 sc[16];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int k = 0; k < 16; k++)
                sc[k] = score_tab[(LZ_CODE_CLASS(LZ_VBYTE(tv, k)) << 5) | LZ_CODE_CLASS(LZ_VBYTE(qv, k))];
            bool go = true;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int k = 0; k < 16; k++) {
                if (go && (u32)k < room) {
                    runr += sc[k];
                    ++sr;
                    if (runr > bestr) { bestr = runr; right_stop = sr; }
                    go = runr >= bestr - xd;
                }
            }
            alive_r = go && ((s32)sr < stopr);
        }
    }
    const u32 extent = (u32)((s32)sr - diag);                   // :2785 (where loop 2 STOPPED)
    if (extent > dend) dend = extent;
    n_bp += (u64)(sr - sl);                                     // :2818
    const s32 sim = bestl + bestr;
    if (sim >= P.min_score) {
        LzHspRec r;
        r.seed_pos1 = pos1; r.seed_pos2 = pos2;
        r.end1 = right_stop; r.length = right_stop - left_start; r.score = sim;
        emit(r);
    }
    return dend;
}
