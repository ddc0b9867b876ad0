// lz_host.cpp -- see lz_host.hpp
#include <string.h>
#include <math.h>
#include <algorithm>
#include <thread>
#include <atomic>
#include <vector>
#include "lz_host.hpp"
#include "lz_lut.hpp"

// strict-seed compilation, restating src/seeds.c:321-640 (parse_one_seed; flips in
// "maintainFlippedBitOrder" order, :603-613) and :1399-1417 (best_shift)
extern "C" int lzgpu_seed_from_pattern(const char* pattern, int with_trans, lz_seed_desc* out)
{
    if (!pattern || !out) return LZGPU_ERR_ARG;
    memset(out, 0, sizeof(*out));
    const char* s = pattern; const char* e = pattern + strlen(pattern);
    while (s < e && (*s == '0' || *s == 'X' || *s == 'x')) s++;
    if (s >= e) return LZGPU_NH_SEED;
    e--;
    while (*e == '0' || *e == 'X' || *e == 'x') e--;
    u64 seed_bits = 0, flip_bits = 0; int length = 0, weight = 0;
    for (const char* c = s; c <= e; c++) {
        if (*c == '1') { seed_bits = (seed_bits << 2) + 3; flip_bits = (flip_bits << 2) + 2; length++; weight += 2; }
        else if (*c == '0' || *c == 'X' || *c == 'x') { seed_bits <<= 2; flip_bits <<= 2; length++; }
        else return LZGPU_NH_SEED;                   // 'T' (half-weight) positions: CPU path
    }
    if (length > 31 || weight > 28 || weight == 0) return LZGPU_NH_SEED;   // src/pos_table.c:1057
    u32 wbits = (u32)((1ull << weight) - 1), covered = (u32)(seed_bits & wbits);
    u64 rem = seed_bits - covered;
    int np = 1;
    out->shift[0] = 0; out->mask[0] = covered;
    while (covered != wbits) {
        u32 uncovered = (~covered) & wbits;
        int best_cov = -1, best = -1, sh = 0;
        for (u64 sb = rem; sb != 0; sb >>= 1, sh++) {
            int cov = __builtin_popcount((u32)(sb & uncovered));
            if (cov > best_cov) { best_cov = cov; best = sh; }
        }
        u32 mask = (u32)(rem >> best) & uncovered;
        covered += mask; rem -= ((u64)mask) << best;
        if (np >= LZGPU_MAX_PARTS) return LZGPU_NH_SEED;
        out->shift[np] = best; out->mask[np] = mask; np++;
    }
    out->num_parts = np; out->length = length; out->weight_bits = weight;
    u32 flips[32]; int nf = 0;
    while (flip_bits != 0) {
        u64 right = flip_bits - (flip_bits & (flip_bits - 1));
        flip_bits -= right;
        u32 packed = 0;
        for (int p = 0; p < np; p++) packed |= (u32)(right >> out->shift[p]) & out->mask[p];
        flips[nf++] = packed;
    }
    int npb = 0;
    out->probe_xor[npb++] = 0;
    if (with_trans == 1) for (int i = 0; i < nf; i++) out->probe_xor[npb++] = flips[i];
    else if (with_trans >= 2)
        for (int i = 0; i < nf; i++) {
            if (npb >= LZGPU_MAX_PROBES) return LZGPU_NH_SEED;
            out->probe_xor[npb++] = flips[i];
            for (int j = i + 1; j < nf; j++) {
                if (npb >= LZGPU_MAX_PROBES) return LZGPU_NH_SEED;
                out->probe_xor[npb++] = flips[i] ^ flips[j];
            }
        }
    out->num_probes = npb;
    return 0;
}

int lzh_seed_to_dev(const lz_seed_desc* sd, LzSeedDev& d)
{
    if (sd->length < 2 || sd->length > 31 || sd->weight_bits < 1 || sd->weight_bits > 28) return LZGPU_NH_SEED;
    if (sd->num_parts < 1 || sd->num_parts > LZ_MAX_PARTS) return LZGPU_NH_SEED;
    if (sd->num_probes < 1 || sd->num_probes > LZ_MAX_PROBES) return LZGPU_NH_SEED;
    memset(&d, 0, sizeof(d));
    d.length = sd->length; d.weight = sd->weight_bits; d.nparts = sd->num_parts; d.nprobes = sd->num_probes;
    for (int i = 0; i < sd->num_parts; i++) { d.shift[i] = sd->shift[i]; d.mask[i] = sd->mask[i]; }
    for (int i = 0; i < sd->num_probes; i++) d.probe_xor[i] = sd->probe_xor[i];
    return 0;
}

void lzh_make_cls(const u8* score_class, const int8_t ctb[256], u8 cls[256])
{
    for (int b = 0; b < 256; b++) {
        u8 v = score_class ? (u8)(score_class[b] & 31) : 0;
        if (ctb[b] < 0) v |= LZ_CODE_INVALID; else v |= (u8)((ctb[b] & 3) << 5);
        cls[b] = v;
    }
}

// Compress the 256x256 matrix to <=32 row classes x <=32 column classes (bytes whose rows /
// columns are identical are interchangeable for scoring).  Exact for any matrix that fits.
int lzh_score_classes(const s32* sub, u8 rowc[256], u8 colc[256], s32 tab[LZ_NCLASS * LZ_NCLASS])
{
    int rep_r[LZ_NCLASS], rep_c[LZ_NCLASS], nr = 0, nc = 0;
    for (int r = 0; r < 256; r++) {
        int k;
        for (k = 0; k < nr; k++) if (memcmp(sub + 256 * r, sub + 256 * rep_r[k], 256 * sizeof(s32)) == 0) break;
        if (k == nr) { if (nr == LZ_NCLASS) return LZGPU_NH_SCORE_CLASSES; rep_r[nr++] = r; }
        rowc[r] = (u8)k;
    }
    std::vector<s32> tr(65536);
    for (int r = 0; r < 256; r++) for (int cc = 0; cc < 256; cc++) tr[256 * cc + r] = sub[256 * r + cc];
    for (int cc = 0; cc < 256; cc++) {
        int k;
        for (k = 0; k < nc; k++) if (memcmp(&tr[256 * cc], &tr[256 * rep_c[k]], 256 * sizeof(s32)) == 0) break;
        if (k == nc) { if (nc == LZ_NCLASS) return LZGPU_NH_SCORE_CLASSES; rep_c[nc++] = cc; }
        colc[cc] = (u8)k;
    }
    for (int i = 0; i < LZ_NCLASS * LZ_NCLASS; i++) tab[i] = 0;
    for (int a = 0; a < nr; a++) for (int b = 0; b < nc; b++) tab[a * LZ_NCLASS + b] = sub[256 * rep_r[a] + rep_c[b]];
    return 0;
}

u32 lzh_small_classes(const u8 rowc[256], const u8 colc[256])
{
    for (int b = 0; b < 256; b++) if (rowc[b] >= 8 || colc[b] >= 8) return 0;
    return 1;
}

// ---- phase-A look-up tables (lz_lut.hpp)
// F[x][w]: the score of a base pair whose Gray codes differ by x, the target's having low bit w
static int lut_classes(const s32 M4[16], s32 F[4][2])
{
    bool have[4][2] = { { false } };
    for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) {
        const int ga = LZ_GRAY(a), gb = LZ_GRAY(b), x = ga ^ gb, w = ga & 1;
        if (have[x][w] && F[x][w] != M4[4 * a + b]) return 0;       // not invariant under complementing both bases
        have[x][w] = true; F[x][w] = M4[4 * a + b];
    }
    return 1;
}

int lzh_lut_eligible(const s32* sub, const int8_t ctb[256], const u8 tocc[256], const u8 qocc[256], s32 xdrop, s32 M4[16])
{
    if (xdrop < 0 || xdrop > 15000) return 0;
    // plain bytes: the score depends on the 2-bit codes only (among the bytes that occur)
    s32 lo = 0;
    for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) {
        bool have = false; s32 v = 0;
        for (int r = 1; r < 256; r++) {
            if (ctb[r] != a) continue;
            for (int c = 1; c < 256; c++) {
                if (ctb[c] != b) continue;
                const bool occ = tocc[r] && qocc[c];
                const s32 x = sub[256 * r + c];
                if (occ) { if (have && x != v) return 0; if (!have) { have = true; v = x; } }
            }
        }
        if (!have) {                                            // the pair never occurs: any representative will do
            for (int r = 1; r < 256 && !have; r++) if (ctb[r] == a) for (int c = 1; c < 256 && !have; c++) if (ctb[c] == b) { v = sub[256 * r + c]; have = true; }
        }
        if (v > 127 || v < -127) return 0;                      // scores travel as signed bytes
        if (v < lo) lo = v;
        M4[4 * a + b] = v;
    }
    s32 F[4][2];
    if (!lut_classes(M4, F)) return 0;
    // (special bytes -- everything outside A, C, G, T -- need no condition: a scan consumes one with its real score, from
    // the class table, whether it ends the scan or not: lz_lut_window)
    // no four-base group loses more than xDrop over the (at most three) bases after a maximum set inside the group
    if (-3 * lo > xdrop) return 0;
    return 1;
}

void lzh_lut_build(const s32 M4[16], s32 xdrop, LzLutEntry* tab)
{
    s32 F[4][2] = { { 0 } };
    lut_classes(M4, F);
    for (int dir = 0; dir < 2; dir++)                           // 0: bases consumed low to high inside a byte (right scans); 1: high to low (left scans)
        for (u32 idx = 0; idx < LZ_LUT_ENTRIES; idx++) {
            const u32 xb = idx & 0xFFu, wn = idx >> 8;
            s32 p = 0, minp = 0x7FFFFFFF, maxp = -0x7FFFFFFF; u32 sc = 0;
            for (u32 k = 0; k < 4; k++) {
                const u32 j = dir == 0 ? k : 3u - k;            // base of the byte consumed k-th
                const s32 v = F[(xb >> (2 * j)) & 3u][(wn >> j) & 1u];
                sc |= (u32)(u8)(int8_t)v << (8 * k);
                p += v;
                if (p < minp) minp = p;
                if (p > maxp) maxp = p;
            }
            const u32 A = minp < 0 ? (u32)(-minp) : 0u;
            const s32 B = xdrop - (maxp > 0 ? maxp : 0);        // the margin never exceeds xDrop
            tab[dir * LZ_LUT_ENTRIES + idx].ab = A | ((u32)(B & 0xFFFF) << 16);
            tab[dir * LZ_LUT_ENTRIES + idx].sc = sc;
        }
}

// src/dna_utilities.c:2888-2936 (compute_entropy with lowerOk == false); same operation order
double lzh_hsp_entropy(const u8* s, const u8* t, int len)
{
    int cA = 0, cC = 0, cG = 0, cT = 0;
    for (int ix = 0; ix < len; ix++) {
        if (s[ix] != t[ix]) continue;
        switch (s[ix]) { case 'A': cA++; break; case 'C': cC++; break; case 'G': cG++; break; case 'T': cT++; break; default: break; }
    }
    return lzh_entropy_from_counts(cA, cC, cG, cT, len);
}

double lzh_entropy_from_counts(int cA, int cC, int cG, int cT, int len)
{
    if (cA + cC + cG + cT < 20) return 1.0;
    double pA = ((double)cA) / ((double)len), pC = ((double)cC) / ((double)len);
    double pG = ((double)cG) / ((double)len), pT = ((double)cT) / ((double)len);
    double qA = (cA != 0) ? log(pA) : 0.0, qC = (cC != 0) ? log(pC) : 0.0;
    double qG = (cG != 0) ? log(pG) : 0.0, qT = (cT != 0) ? log(pT) : 0.0;
    return -(pA * qA + pC * qC + pG * qG + pT * qT) / log(4.0);
}

static bool host_window_word(const u8* seq, u32 pos, const LzSeedDev& sd, const int8_t* ctb, u32& packed)
{
    u64 w = 0;
    for (int k = 0; k < sd.length; k++) {
        int b = ctb[seq[pos - sd.length + k]];
        if (b < 0) return false;
        w = (w << 2) | (u64)b;
    }
    packed = lz_apply_seed(sd, w);
    return true;
}

// discovery order of the candidates: query position up, probe order, target position down
// (src/seed_search.c:506-533, :832) -- one 64-bit key (pos2 | probe | ~pos1's upper bits would not fit, so
// two words) compared lexicographically
struct RecKey { u64 hi; u32 lo, idx; };       // hi = pos2 << 32 | probe, lo = ~pos1

int lzh_finish_hsps(const LzHspRec* recs, u32 n_rec, const u8* thost, const u8* qhost,
                    const LzSeedDev& sd, const int8_t ctb[256], s32 K, int entropic,
                    std::vector<lz_hsp>& out, const u32* match_counts, std::vector<u64>* order_out)
{
    out.clear();
    if (order_out) order_out->clear();
    // The GPU is idle while this runs (the call is synchronous).  Up to sixteen threads of a pool that lives as long as the
    // library go through four passes: keys + a count per (part, range of query positions); the keys dealt out into the ranges
    // and each range sorted on its own (the order is by query position first: ranges, not merges -- the last level of a merge
    // tree is one thread merging everything); entropy factors and survivors counted; output.
    // (Round 2: four threads started anew for every phase, 3.8 ms per search of the 50 Mbp pair; round 3: up to sixteen
    // started once per call, chunk sorts + three levels of merges, 2.2-2.8 ms.)
    auto less = [](const RecKey& x, const RecKey& y) { return x.hi != y.hi ? x.hi < y.hi : x.lo < y.lo; };
    static ForkJoin pool([]() { const u32 hw = std::thread::hardware_concurrency(); return hw >= 32 ? 15u : hw >= 16 ? 7u : hw >= 8 ? 3u : hw >= 4 ? 1u : 0u; }());
    static std::mutex pool_m;                                   // (one search at a time uses it)
    std::unique_lock<std::mutex> pool_lock(pool_m, std::defer_lock);
    u32 T = 1;
    if (n_rec >= 16384) { pool_lock.lock(); T = pool.threads() + 1u; }
    auto part = [&](u32 t) { return (u32)((u64)n_rec * t / T); };
    const s32 zero_thresh = K > 0 ? K : 0;                      // src/lastz.c:2937-2939
    u32 max_p2 = 0;
    for (u32 i = 0; i < n_rec; i++) if (recs[i].seed_pos2 > max_p2) max_p2 = recs[i].seed_pos2;
    const u32 R = T;                                            // ranges of query positions, about n_rec / T records each on even data
    auto range_of = [&](u32 p2) { return (u32)((u64)p2 * R / ((u64)max_p2 + 1)); };
    std::vector<RecKey> keys(n_rec), order(n_rec);
    std::vector<u32> cnt((size_t)T * R, 0), start((size_t)T * R + 1, 0);
    std::vector<s32> sims(n_rec);
    std::vector<u32> kept((size_t)T + 1, 0);
    std::atomic<int> bad{0};
    const std::function<void(size_t, size_t)> make_keys = [&](size_t lo, size_t hi) {
        for (size_t t = lo; t < hi; t++)
            for (u32 i = part((u32)t); i < part((u32)t + 1); i++) {
                u32 pt = 0, pq = 0, probe = 0;
                if (match_counts) probe = match_counts[5 * (size_t)i + 4];
                else {
                    if (!host_window_word(thost, recs[i].seed_pos1, sd, ctb, pt) ||
                        !host_window_word(qhost, recs[i].seed_pos2, sd, ctb, pq)) { bad.store(1); probe = 0; }
                    else { const u32 x = pt ^ pq; for (probe = 0; probe < (u32)sd.nprobes; probe++) if (sd.probe_xor[probe] == x) break; }
                }
                if (probe >= (u32)sd.nprobes) { bad.store(1); probe = 0; }
                keys[i] = { ((u64)recs[i].seed_pos2 << 32) | probe, ~recs[i].seed_pos1, i };
                cnt[t * R + range_of(recs[i].seed_pos2)]++;
            }
    };
    const std::function<void(size_t, size_t)> deal = [&](size_t lo, size_t hi) {      // part t's keys to their ranges
        for (size_t t = lo; t < hi; t++) {
            std::vector<u32> at(R);
            for (u32 r = 0; r < R; r++) at[r] = start[(size_t)r * T + t];
            for (u32 i = part((u32)t); i < part((u32)t + 1); i++) order[at[range_of((u32)(keys[i].hi >> 32))]++] = keys[i];
        }
    };
    const std::function<void(size_t, size_t)> sort_range = [&](size_t lo, size_t hi) {
        for (size_t r = lo; r < hi; r++) std::sort(order.begin() + start[r * T], order.begin() + start[(r + 1) * T], less);
    };
    const std::function<void(size_t, size_t)> entropy = [&](size_t lo, size_t hi) {
        for (size_t t = lo; t < hi; t++) {
            u32 n_kept = 0;
            for (u32 k = part((u32)t); k < part((u32)t + 1); k++) {
                const LzHspRec& r = recs[order[k].idx];
                s32 diag = (s32)r.seed_pos1 - (s32)r.seed_pos2;
                u32 pos1 = r.end1, pos2 = (u32)((s32)pos1 - diag), length = r.length;
                s32 sim = r.score;
                if (entropic && sim >= zero_thresh && (s64)sim <= 3 * (s64)K) {
                    const u32* mc = match_counts ? match_counts + 5 * (size_t)order[k].idx : nullptr;
                    double q = mc ? lzh_entropy_from_counts((int)mc[0], (int)mc[1], (int)mc[2], (int)mc[3], (int)length)
                                  : lzh_hsp_entropy(thost + pos1 - length, qhost + pos2 - length, (int)length);
                    sim = (s32)(sim * q);                       // "similarity *= q" on an s32 score
                }
                sims[k] = sim;
                if (sim >= K) n_kept++;
            }
            kept[t + 1] = n_kept;
        }
    };
    const std::function<void(size_t, size_t)> write_out = [&](size_t lo, size_t hi) {
        for (size_t t = lo; t < hi; t++) {
            u32 w = kept[t];
            for (u32 k = part((u32)t); k < part((u32)t + 1); k++) {
                if (sims[k] < K) continue;
                const LzHspRec& r = recs[order[k].idx];
                const s32 diag = (s32)r.seed_pos1 - (s32)r.seed_pos2;
                out[w] = { r.end1, (u32)((s32)r.end1 - diag), r.length, sims[k] };
                if (order_out) { (*order_out)[2 * (size_t)w] = order[k].hi; (*order_out)[2 * (size_t)w + 1] = (u64)order[k].lo; }
                w++;
            }
        }
    };
    if (T > 1) pool.begin_burst();
    pool.run(T, 1, make_keys);
    if (bad.load()) { if (T > 1) pool.end_burst(); return LZGPU_ERR_STATE; }
    { u32 acc = 0; for (u32 r = 0; r < R; r++) for (u32 t = 0; t < T; t++) { start[(size_t)r * T + t] = acc; acc += cnt[(size_t)t * R + r]; } start[(size_t)R * T] = acc; }
    pool.run(T, 1, deal);
    pool.run(R, 1, sort_range);
    pool.run(T, 1, entropy);
    for (u32 t = 0; t < T; t++) kept[t + 1] += kept[t];
    out.resize(kept[T]);
    if (order_out) order_out->resize(2 * (size_t)kept[T]);
    pool.run(T, 1, write_out);
    if (T > 1) { pool.end_burst(); pool_lock.unlock(); }
    return 0;
}
