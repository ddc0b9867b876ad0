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
LZ_HD u32 lz_count_hits_at(const u8* qcode, u32 pos2, u32 lo, const LzSeedDev& sd, const u32* wstart, bool& valid)
{
    u32 packed;
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
};

// One bucket (= one value of hashedDiag) of the diagonal hash: process its hits of this chunk
// in enumeration order.  This is process_for_simple_hit + xdrop_extend_seed_hit
// (src/seed_search.c:1056-1192, 2528-2959) with diagEnd[h] held in a register.
//   keys[i0..i1)  this bucket's hits, already in discovery order
//   dend          diagEnd[h] on entry (0 == inactive, src/seed_search.c:1097-1111)
// Returns the updated diagEnd[h].  emit(rec) is called for every extension scoring >= min_score.
template <class Emit>
LZ_HD u32 lz_extend_bucket(const LzExtendParams& P, const s32* score_tab /*[32*32]*/,
                           const u64* keys, u32 i0, u32 i1, u32 dend,
                           u64& n_ext, u64& n_bp, Emit&& emit)
{
    const u32 L = P.seed_len;
    for (u32 i = i0; i < i1; i++) {
        const u64 key = keys[i];
        const u32 pos2 = (u32)key;
        const s32 diag = (s32)(u32)(key >> 32);
        const u32 pos1 = pos2 + (u32)diag;
        if (dend > pos2 - L) continue;                          // :1113

        n_ext++;
        // left extension, :2612-2632
        s32 stopl = (s32)dend + diag;  if (stopl < 0) stopl = 0;
        u32 s1 = pos1, left_start = pos1;
        s32 run = 0, left = 0;
        {
            const u8* tp = P.tcode; const u8* qp = P.qcode;
            while ((s32)s1 > stopl && run >= left - P.xdrop) {
                --s1;
                run += score_tab[(LZ_CODE_CLASS(tp[s1]) << 5) | LZ_CODE_CLASS(qp[(s32)s1 - diag])];
                if (run > left) { left = run; left_start = s1; }
            }
        }
        const u32 left_block = s1;
        // right extension, :2675-2694
        s32 stopr = ((s32)P.tlen <= (s32)P.qlen + diag) ? (s32)P.tlen : (s32)P.qlen + diag;
        s1 = pos1; u32 right_stop = pos1;
        s32 right = 0; run = 0;
        {
            const u8* tp = P.tcode; const u8* qp = P.qcode;
            while ((s32)s1 < stopr && run >= right - P.xdrop) {
                run += score_tab[(LZ_CODE_CLASS(tp[s1]) << 5) | LZ_CODE_CLASS(qp[(s32)s1 - diag])];
                s1++;
                if (run > right) { right = run; right_stop = s1; }
            }
        }
        const u32 extent = (u32)((s32)s1 - diag);               // :2785
        if (extent > dend) dend = extent;
        n_bp += (u64)(s1 - left_block);                         // :2818

        const s32 sim = left + right;
        if (sim >= P.min_score) {
            LzHspRec r;
            r.seed_pos1 = pos1; r.seed_pos2 = pos2;
            r.end1 = right_stop; r.length = right_stop - left_start; r.score = sim;
            emit(r);
        }
    }
    return dend;
}
