// emul_seed.cpp -- TEST INFRASTRUCTURE: runs the seed stage's per-lane device logic
// (lastz_amd/csrc/lz_common.hpp) and its host pieces (lz_host.cpp) serially on the CPU, in
// exactly the GPU pipeline's decomposition:
//   encode -> table (pairs in descending position, stable sort by word, CSR bounds)
//   count -> exclusive scan -> chunk plan -> per chunk: fill -> stable partition by hash bucket
//   -> bucket bounds -> one "lane" per bucket -> host finish.
// It exists so that "-m 'not gpu'" tests can check the decomposition against the oracle on a
// machine without a GPU.  It is never linked into liblzgpu.so and is not a fallback.
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../lastz_amd/csrc/lz_common.hpp"
#include "../../lastz_amd/csrc/lz_host.hpp"
#include "../../lastz_amd/csrc/lz_lut.hpp"
#include "../../lastz_amd/csrc/lz_coop.hpp"

struct Emul {
    std::vector<u8> traw, tcode;   // with LZ_SEQ_PAD either side
    u32 tlen = 0, start = 0, end = 0, step = 1;
    LzSeedDev sd;
    int8_t ctb[256];
    std::vector<u32> wstart, wpos;
    u64 hit_cap = 1ull << 28;
    lz_counters cnt = {};
    int last_scan_mode = -1;
};
static Emul E;

// phase B's slow path as the device runs it: the wave-cooperative extension (lz_coop.hpp) instead of the serial
// loops of lz_reextend; same contract
template <class Emit>
static u32 reextend_coop(const LzExtendParams& P, const s32* tab, u32 pos2, s32 diag, u32 dend, u64& n_bp, Emit&& emit)
{
    const u32 pos1 = pos2 + (u32)diag;
    s32 stopl = (s32)dend + diag; if (stopl < 0) stopl = 0;
    const s32 stopr = ((s32)P.tlen <= (s32)P.qlen + diag) ? (s32)P.tlen : (s32)P.qlen + diag;
    auto score = [&](u32 i) { return tab[(LZ_CODE_CLASS(P.tcode[i]) << 5) | LZ_CODE_CLASS(P.qcode[(s32)i - diag])]; };
    const LzCoopSide L = lz_coop_scan_host<false>(pos1, stopl, P.xdrop, score);
    const LzCoopSide R = lz_coop_scan_host<true>(pos1, stopr, P.xdrop, score);
    const u32 extent = (u32)((s32)R.stop_pos - diag);
    if (extent > dend) dend = extent;
    n_bp += (u64)(R.stop_pos - L.stop_pos);
    const s32 sim = L.best + R.best;
    if (sim >= P.min_score) { LzHspRec r; r.seed_pos1 = pos1; r.seed_pos2 = pos2; r.end1 = R.best_pos; r.length = R.best_pos - L.best_pos; r.score = sim; emit(r); }
    return dend;
}
static u64 g_rx[4];   // profiling aid: re-extensions phase B runs, bases they scan, SLOW records, records

static void encode(const std::vector<u8>& raw, std::vector<u8>& code, const u8 cls[256])
{
    code.assign(raw.size(), LZ_CODE_INVALID);
    for (size_t i = LZ_SEQ_PAD; i + LZ_SEQ_PAD < raw.size() + 0; i++) code[i] = cls[raw[i]];
}

extern "C" int emul_set_hit_capacity(uint64_t n) { E.hit_cap = n; return 0; }
extern "C" void emul_counters_reset() { memset(&E.cnt, 0, sizeof(E.cnt)); }
extern "C" void emul_counters_get(lz_counters* o) { *o = E.cnt; }

extern "C" int emul_table_prepare(const u8* t, u32 tlen, u32 start, u32 end, const int8_t* ctb,
                                  const lz_seed_desc* seed, u32 step)
{
    if (end == 0) end = tlen;
    if (end <= start || end > tlen || step < 1) return LZGPU_ERR_ARG;
    int rc = lzh_seed_to_dev(seed, E.sd); if (rc) return rc;
    E.tlen = tlen; E.start = start; E.end = end; E.step = step; memcpy(E.ctb, ctb, 256);
    E.traw.assign((size_t)tlen + 2 * LZ_SEQ_PAD, 0);
    memcpy(E.traw.data() + LZ_SEQ_PAD, t, tlen);
    u8 cls[256]; lzh_make_cls(nullptr, ctb, cls);
    encode(E.traw, E.tcode, cls);
    const u8* tc = E.tcode.data() + LZ_SEQ_PAD;
    // k_table_words
    u32 n = end - start, nwords = 1u << E.sd.weight;
    std::vector<std::pair<u32, u32>> kv(n);
    for (u32 j = 0; j < n; j++) {
        u32 p = end - j, key = nwords, packed;
        if (p >= start + (u32)E.sd.length && p % step == 0 && lz_window_word(tc, p, E.sd, packed)) key = packed;
        kv[j] = { key, p };
    }
    std::stable_sort(kv.begin(), kv.end(), [](auto& a, auto& b) { return a.first < b.first; });
    // k_key_bounds_u32
    E.wstart.assign((size_t)nwords + 1, 0);
    for (u32 i = 0; i <= n; i++) {
        s64 kp = (i == 0) ? -1 : (s64)kv[i - 1].first, k = (i == n) ? (s64)nwords : (s64)kv[i].first;
        if (k > nwords) k = nwords;
        if (kp > nwords) kp = nwords;
        for (s64 w = kp + 1; w <= k; w++) E.wstart[w] = i;
    }
    u32 nw = E.wstart[nwords];
    E.wpos.resize(nw);
    for (u32 i = 0; i < nw; i++) E.wpos[i] = kv[i].second;
    return 0;
}

extern "C" u64 emul_table_csr(u32* wstart, u32* wpos)
{
    if (wstart) memcpy(wstart, E.wstart.data(), E.wstart.size() * 4);
    if (wpos) memcpy(wpos, E.wpos.data(), E.wpos.size() * 4);
    return E.wpos.size();
}

extern "C" int emul_seed_hit_search(const lz_search_args* a, lz_hsp** out, uint64_t* n_out)
{
    *out = nullptr; *n_out = 0;
    u8 rowc[256], colc[256], cls[256]; s32 tab[LZ_NCLASS * LZ_NCLASS];
    int rc = lzh_score_classes(a->sub, rowc, colc, tab); if (rc) return rc;
    lzh_make_cls(rowc, E.ctb, cls); encode(E.traw, E.tcode, cls);
    const u32 qlen = a->qlen;
    std::vector<u8> qraw((size_t)qlen + 2 * LZ_SEQ_PAD, 0), qcode;
    memcpy(qraw.data() + LZ_SEQ_PAD, a->query, qlen);
    lzh_make_cls(colc, E.ctb, cls); encode(qraw, qcode, cls);
    const u8* tc = E.tcode.data() + LZ_SEQ_PAD; const u8* qc = qcode.data() + LZ_SEQ_PAD;
    u32 lo = a->start, hi = a->end ? a->end : qlen;
    if (hi <= lo || hi > qlen) return LZGPU_ERR_ARG;
    const u32 L = E.sd.length;
    if (qlen < L) return 0;
    const u32 n = hi - lo;
    std::vector<u32> cnt(n); std::vector<u64> off(n + 1);
    u64 words = 0;
    for (u32 i = 0; i < n; i++) { bool v; u32 pkd; cnt[i] = lz_count_hits_at(qc, lo + i + 1, lo, E.sd, E.wstart.data(), v, pkd); words += v; }
    off[0] = 0; for (u32 i = 0; i < n; i++) off[i + 1] = off[i] + cnt[i];
    std::vector<LzChunk> chunks;
    rc = lzh_plan_chunks(n, E.hit_cap, 4096, [&](u32 i) { return off[i > n ? n : i]; }, chunks);
    if (rc) return rc;
    std::vector<u32> diag_end(LZ_DIAG_SIZE, 0), bstart(LZ_DIAG_SIZE + 1);
    std::vector<LzHspRec> recs; std::vector<lz_hsp> plain;
    LzExtendParams P; P.tcode = tc; P.tlen = E.tlen; P.qcode = qc; P.qlen = qlen;
    P.xdrop = a->xdrop; P.min_score = a->hsp_threshold; P.seed_len = L;
    P.cls8 = lzh_small_classes(rowc, colc);
    // 4-bit codes as the device builds them (k_pack_nibbles), with slack for whole 16-byte loads
    auto nibbles = [](const std::vector<u8>& code) {
        std::vector<u8> nb(code.size() / 2 + 80, 0);
        for (size_t b = 0; 2 * b + 1 < code.size(); b++) nb[b] = (u8)((code[2 * b] & 7u) | ((code[2 * b + 1] & 7u) << 4));
        return nb;
    };
    std::vector<u8> tnib, qnib;
    const bool use_nib = P.cls8 && !getenv("EMUL_NO_NIBBLES");
    if (use_nib) { tnib = nibbles(E.tcode); qnib = nibbles(qcode); }
    P.tnib = use_nib ? tnib.data() : nullptr; P.qnib = use_nib ? qnib.data() : nullptr;
    s32 tab8[64];
    for (int k = 0; k < 64; k++) tab8[k] = tab[(k >> 3) * LZ_NCLASS + (k & 7)];
    u64 n_ext = 0, n_bp = 0;
    // 2-bit codes + special masks as k_pack2 builds them, occurrence sets as k_byte_presence does
    auto pack2 = [](const std::vector<u8>& code /*with LZ_SEQ_PAD*/, u32 len, std::vector<u8>& two, std::vector<u8>& spc) {
        const size_t nmask = ((size_t)len + 2 * LZ_PAD2 + 7) / 8 + 16;
        two.assign(nmask * 2 + 32, 0); spc.assign(nmask + 32, 0xFF);
        for (size_t j = 0; j < nmask; j++) {
            u32 bits = 0, m = 0;
            for (int k = 0; k < 8; k++) {
                const s64 i = (s64)j * 8 - LZ_PAD2 + k;
                const u32 c = (i >= -(s64)LZ_SEQ_PAD && i < (s64)len + LZ_SEQ_PAD) ? code[(size_t)(i + LZ_SEQ_PAD)] : (u32)LZ_CODE_INVALID;
                if (c & LZ_CODE_INVALID) m |= 1u << k; else bits |= LZ_GRAY(LZ_CODE_BITS(c)) << (2 * k);
            }
            spc[j] = (u8)m; two[2 * j] = (u8)bits; two[2 * j + 1] = (u8)(bits >> 8);
        }
    };
    std::vector<u8> t2, tsp, q2, qsp;
    pack2(E.tcode, E.tlen, t2, tsp); pack2(qcode, qlen, q2, qsp);
    u8 tocc[256] = { 0 }, qocc[256] = { 0 }; bool tspecial = false, qspecial = false;
    for (u32 i = 0; i < E.tlen; i++) { const u8 b = E.traw[LZ_SEQ_PAD + i]; tocc[b] = 1; if (E.ctb[b] < 0) tspecial = true; }
    for (u32 i = 0; i < qlen; i++) { const u8 b = a->query[i]; qocc[b] = 1; if (E.ctb[b] < 0) qspecial = true; }
    s32 M4[16] = { 0 };
    int scan_mode = 2;
    if (lzh_lut_eligible(a->sub, E.ctb, tocc, qocc, a->xdrop, M4)) scan_mode = (tspecial || qspecial) ? 1 : 0;
    if (const char* f = getenv("EMUL_SCAN_MODE")) { const int v = atoi(f); if (scan_mode < 2 && v > scan_mode) scan_mode = v > 2 ? 2 : v; }
    E.last_scan_mode = scan_mode;
    std::vector<LzLutEntry> lut(2 * LZ_LUT_ENTRIES);
    if (scan_mode < 2) lzh_lut_build(M4, a->xdrop, lut.data());
    LzLutParams Q; Q.t2 = t2.data(); Q.q2 = q2.data(); Q.tsp = tsp.data(); Q.qsp = qsp.data(); Q.xdrop = a->xdrop;
    Q.t2x = nullptr; Q.tspx = nullptr; Q.tcode = tc; Q.qcode = qc;
    for (auto& ch : chunks) {
        std::vector<u64> keys(ch.nh);
        for (u32 i = ch.i0; i < ch.i1; i++)
            if (cnt[i]) lz_fill_hits_at(qc, lo + i + 1, E.sd, E.wstart.data(), E.wpos.data(), keys.data() + (off[i] - ch.base));
        if (!a->extend) { for (u64 k : keys) { u32 p2 = (u32)k; plain.push_back({ p2 + (u32)(k >> 32), p2, L, 0 }); } continue; }
        // phase A on the hits in discovery order (look-up-table scans on 2-bit codes when the matrix allows it,
        // k_scan_hits<0/1>, else the byte-code scans, <2>), 8-byte records, stable partition by the high 8 hash
        // bits; phase B (k_settle): per partition, tiles of LZ_ST_TILE records are dealt out to the 256 buckets
        // and every bucket walks its list in tile order
        std::vector<u64> rec(keys.size());
        for (size_t i = 0; i < keys.size(); i++) {
            u32 sm;
            if (scan_mode == 0)      sm = lz_lut_probe_hit<false>(Q, lut.data(), P.tlen, P.qlen, P.min_score, keys[i]);
            else if (scan_mode == 1) sm = lz_lut_probe_hit<true>(Q, lut.data(), P.tlen, P.qlen, P.min_score, keys[i], tab);
            else                     sm = lz_probe_hit(P, tab, tab8, P.cls8 != 0, keys[i]);
            rec[i] = lz_hit_record(keys[i], sm);
        }
        std::vector<size_t> ord(keys.size());
        for (size_t i = 0; i < ord.size(); i++) ord[i] = i;
        std::stable_sort(ord.begin(), ord.end(), [&](size_t x, size_t y) { return ((keys[x] >> 40) & 0xFF) < ((keys[y] >> 40) & 0xFF); });
        size_t p0 = 0;
        for (u32 bin = 0; bin < 256; bin++) {
            size_t p1 = p0;
            while (p1 < ord.size() && ((keys[ord[p1]] >> 40) & 0xFF) == bin) p1++;
            for (size_t t0 = p0; t0 < p1; t0 += 8192) {
                const size_t t1 = std::min(p1, t0 + 8192);
                std::vector<std::vector<u64>> lists(256);
                for (size_t k = t0; k < t1; k++) lists[LZ_REC_LOW8(rec[ord[k]])].push_back(rec[ord[k]]);
                for (u32 b = 0; b < 256; b++) {
                    const u32 h = bin * 256 + b;
                    for (u64 r : lists[b]) {
                        const u64 bp0 = n_bp, ex0 = n_ext;
                        if (LZ_REC_SLOW(r) && !(diag_end[h] > LZ_REC_POS2(r) - P.seed_len) && !getenv("EMUL_SERIAL_REEXTEND")) {
                            n_ext++;
                            diag_end[h] = reextend_coop(P, tab, LZ_REC_POS2(r), (s32)((LZ_REC_PAYLOAD(r) << 16) | h), diag_end[h], n_bp,
                                                        [&](const LzHspRec& x) { recs.push_back(x); });
                        } else
                        lz_settle_record(P, tab, r, h, diag_end[h], n_ext, n_bp, [&](const LzHspRec& x) { recs.push_back(x); });
                        g_rx[3]++; if (LZ_REC_SLOW(r)) { g_rx[2]++; if (n_ext != ex0) { g_rx[0]++; g_rx[1] += n_bp - bp0; } }
                    }
                }
            }
            p0 = p1;
        }
    }
    E.cnt.words += words; E.cnt.raw_hits += off[n]; E.cnt.extensions += n_ext; E.cnt.bp_extended += n_bp;
    std::vector<lz_hsp> fin;
    if (!a->extend) fin = plain;
    else {
        // candidates arrive in arbitrary (atomic) order on the GPU: shuffle deterministically here
        std::reverse(recs.begin(), recs.end());
        rc = lzh_finish_hsps(recs.data(), (u32)recs.size(), E.traw.data() + LZ_SEQ_PAD, a->query, E.sd, E.ctb,
                             a->hsp_threshold, a->entropic, fin, nullptr, nullptr);
        if (rc) return rc;
        E.cnt.hsps += fin.size();
    }
    lz_hsp* res = (lz_hsp*)malloc((fin.size() ? fin.size() : 1) * sizeof(lz_hsp));
    if (!fin.empty()) memcpy(res, fin.data(), fin.size() * sizeof(lz_hsp));
    *out = res; *n_out = fin.size();
    return 0;
}
extern "C" void emul_free(void* p) { free(p); }
extern "C" void emul_reext_stats(u64* o) { for (int k = 0; k < 4; k++) o[k] = g_rx[k]; }
extern "C" int emul_last_scan_mode() { return E.last_scan_mode; }

// Self-test of the three block scanners on random class codes: the masked general scan
// (lz_scan_left16/right16 without the 8x8 table), the whole-block byte scan and the 4-bit scan must agree
// on (bases consumed, run, best) for every block, both directions.  Returns the number of mismatches.
extern "C" int emul_scan_selftest(uint32_t seed, uint32_t rounds)
{
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x >> 11); };
    s32 tab[LZ_NCLASS * LZ_NCLASS], tab8[64];
    for (int i = 0; i < LZ_NCLASS * LZ_NCLASS; i++) tab[i] = (s32)(rnd() % 400) - 300;
    for (int k = 0; k < 64; k++) tab8[k] = tab[(k >> 3) * LZ_NCLASS + (k & 7)];
    int bad = 0;
    for (uint32_t r = 0; r < rounds; r++) {
        std::vector<u8> t(256, 0), q(256, 0);
        for (auto& b : t) b = (u8)((rnd() % 8) | ((rnd() % 4) << 5));
        for (auto& b : q) b = (u8)((rnd() % 8) | ((rnd() % 4) << 5));
        std::vector<u8> tn(200, 0), qn(200, 0);                 // LZ_SEQ_PAD nibbles of padding in front
        for (size_t b = 0; b < 128; b++) { tn[LZ_SEQ_PAD / 2 + b] = (u8)((t[2 * b] & 7) | ((t[2 * b + 1] & 7) << 4)); qn[LZ_SEQ_PAD / 2 + b] = (u8)((q[2 * b] & 7) | ((q[2 * b + 1] & 7) << 4)); }
        const s32 xd = (s32)(rnd() % 900) + 10;
        const u32 pos = 64 + rnd() % 100;                       // block [pos-16,pos) to the left, [pos,pos+16) to the right
        for (int right = 0; right < 2; right++) {
            const s32 run0 = (s32)(rnd() % 500) - 100, best0 = run0 + (s32)(rnd() % 300);
            u32 s1 = pos, s2 = pos; s32 r1 = run0, b1 = best0, r2 = run0, b2 = best0, r3 = run0, b3 = best0;
            const LzVec16 tv = lz_load16(t.data() + (right ? pos : pos - 16)), qv = lz_load16(q.data() + (right ? pos : pos - 16));
            bool a1, a2;
            if (right) { a1 = lz_scan_right16(tab, tab8, false, xd, tv, qv, 100000, s1, r1, b1); a2 = lz_scan_right16(tab, tab8, true, xd, tv, qv, 100000, s2, r2, b2); }
            else       { a1 = lz_scan_left16(tab, tab8, false, xd, tv, qv, -100000, s1, r1, b1); a2 = lz_scan_left16(tab, tab8, true, xd, tv, qv, -100000, s2, r2, b2); }
            u32 nt[3], nq[3];
            lz_load_nib<1>(tn.data(), right ? (s64)pos : (s64)pos - 16, nt);
            lz_load_nib<1>(qn.data(), right ? (s64)pos : (s64)pos - 16, nq);
            const u32 nok = right ? lz_scan16_nib<false>(tab8, xd, nt, nq, r3, b3) : lz_scan16_nib<true>(tab8, xd, nt, nq, r3, b3);
            const u32 c3 = nok < 16 ? nok + 1 : 16, c1 = right ? s1 - pos : pos - s1, c2 = right ? s2 - pos : pos - s2;
            if (c1 != c2 || c1 != c3 || b1 != b2 || b1 != b3 || a1 != a2 || a1 != (nok == 16)) bad++;
            if (a1 && (r1 != r2 || r1 != r3)) bad++;            // run only matters while the scan goes on
        }
    }
    return bad;
}

// lzh_sort4 (four-thread sort used by the host phases) against std::sort
extern "C" int emul_sort4_selftest(uint32_t seed, uint32_t n)
{
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 7;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    std::vector<std::pair<uint64_t, uint32_t>> a(n), b;
    for (uint32_t i = 0; i < n; i++) a[i] = { rnd() % (n / 3 + 1), (uint32_t)(rnd() % 5) };   // many ties on the first key
    b = a;
    auto less = [](const std::pair<uint64_t, uint32_t>& p, const std::pair<uint64_t, uint32_t>& q) { return p < q; };
    lzh_sort4(a.begin(), a.end(), less);
    std::sort(b.begin(), b.end(), less);
    return a == b ? 0 : 1;
}

// lzh_finish_hsps with many candidates (the multi-threaded path: chunk sorts, merge levels, entropy factors behind
// spin barriers) against a plain serial statement of the same thing.  Returns the number of differences.
extern "C" int emul_finish_selftest(uint32_t seed, uint32_t n)
{
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 11;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x >> 9); };
    LzSeedDev sd; memset(&sd, 0, sizeof(sd)); sd.length = 19; sd.weight = 24; sd.nprobes = 13;
    std::vector<LzHspRec> recs(n); std::vector<u32> mc((size_t)n * 5);
    for (uint32_t i = 0; i < n; i++) {
        LzHspRec& r = recs[i];
        r.seed_pos2 = 100 + rnd() % (n / 4 + 50);              // many candidates per query position
        r.seed_pos1 = 100 + rnd() % 100000; r.length = 30 + rnd() % 300; r.end1 = r.seed_pos1 + rnd() % 100; r.score = 2500 + (s32)(rnd() % 9000);
        u32 left = r.length;
        for (int k = 0; k < 4; k++) { const u32 c = rnd() % (left / 2 + 1); mc[5 * (size_t)i + k] = c; left -= c; }
        mc[5 * (size_t)i + 4] = rnd() % 13;
    }
    int8_t ctb[256]; memset(ctb, -1, 256);
    std::vector<lz_hsp> got; std::vector<u64> ord;
    if (lzh_finish_hsps(recs.data(), n, nullptr, nullptr, sd, ctb, 3000, 1, got, mc.data(), &ord)) return -1;
    // serial reference
    std::vector<u32> ix(n); for (uint32_t i = 0; i < n; i++) ix[i] = i;
    auto key = [&](u32 i) { return std::make_tuple(recs[i].seed_pos2, mc[5 * (size_t)i + 4], ~recs[i].seed_pos1); };
    std::stable_sort(ix.begin(), ix.end(), [&](u32 a, u32 b) { return key(a) < key(b); });
    std::vector<lz_hsp> want;
    for (u32 i : ix) {
        const LzHspRec& r = recs[i]; s32 sim = r.score;
        if (sim >= 3000 && (s64)sim <= 9000) sim = (s32)(sim * lzh_entropy_from_counts((int)mc[5 * (size_t)i], (int)mc[5 * (size_t)i + 1], (int)mc[5 * (size_t)i + 2], (int)mc[5 * (size_t)i + 3], (int)r.length));
        if (sim < 3000) continue;
        const s32 diag = (s32)r.seed_pos1 - (s32)r.seed_pos2;
        want.push_back({ r.end1, (u32)((s32)r.end1 - diag), r.length, sim });
    }
    if (want.size() != got.size()) return 1 + (int)want.size();
    int bad = 0;
    for (size_t k = 0; k < want.size(); k++) if (memcmp(&want[k], &got[k], sizeof(lz_hsp)) != 0) bad++;
    return bad;
}
