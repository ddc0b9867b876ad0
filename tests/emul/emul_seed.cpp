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

struct Emul {
    std::vector<u8> traw, tcode;   // with LZ_SEQ_PAD either side
    u32 tlen = 0, start = 0, end = 0, step = 1;
    LzSeedDev sd;
    int8_t ctb[256];
    std::vector<u32> wstart, wpos;
    u64 hit_cap = 1ull << 28;
    lz_counters cnt = {};
};
static Emul E;

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
    for (auto& ch : chunks) {
        std::vector<u64> keys(ch.nh);
        for (u32 i = ch.i0; i < ch.i1; i++)
            if (cnt[i]) lz_fill_hits_at(qc, lo + i + 1, E.sd, E.wstart.data(), E.wpos.data(), keys.data() + (off[i] - ch.base));
        if (!a->extend) { for (u64 k : keys) { u32 p2 = (u32)k; plain.push_back({ p2 + (u32)(k >> 32), p2, L, 0 }); } continue; }
        // phase A on the unsorted hits, then the (key, summary) pairs are partitioned together
        std::vector<std::pair<u64, u32>> kv(keys.size());
        for (size_t i = 0; i < keys.size(); i++) kv[i] = { keys[i], lz_probe_hit(P, tab, P.cls8 ? tab8 : nullptr, keys[i]) };
        std::stable_sort(kv.begin(), kv.end(), [](auto& x, auto& y) { return ((x.first >> 32) & 0xFFFF) < ((y.first >> 32) & 0xFFFF); });
        std::vector<u32> summ(keys.size());
        for (size_t i = 0; i < keys.size(); i++) { keys[i] = kv[i].first; summ[i] = kv[i].second; }
        u64 nk = keys.size();
        for (u64 i = 0; i <= nk; i++) {
            s32 bp = (i == 0) ? -1 : (s32)((keys[i - 1] >> 32) & 0xFFFF), b = (i == nk) ? (s32)LZ_DIAG_SIZE : (s32)((keys[i] >> 32) & 0xFFFF);
            for (s32 w = bp + 1; w <= b; w++) bstart[w] = (u32)i;
        }
        for (u32 h = 0; h < LZ_DIAG_SIZE; h++) {
            if (bstart[h] == bstart[h + 1]) continue;
            diag_end[h] = lz_extend_bucket(P, tab, keys.data(), summ.data(), bstart[h], bstart[h + 1], diag_end[h], n_ext, n_bp,
                                           [&](const LzHspRec& r) { recs.push_back(r); });
        }
    }
    E.cnt.words += words; E.cnt.raw_hits += off[n]; E.cnt.extensions += n_ext; E.cnt.bp_extended += n_bp;
    std::vector<lz_hsp> fin;
    if (!a->extend) fin = plain;
    else {
        // candidates arrive in arbitrary (atomic) order on the GPU: shuffle deterministically here
        std::reverse(recs.begin(), recs.end());
        rc = lzh_finish_hsps(recs.data(), (u32)recs.size(), E.traw.data() + LZ_SEQ_PAD, a->query, E.sd, E.ctb,
                             a->hsp_threshold, a->entropic, fin);
        if (rc) return rc;
        E.cnt.hsps += fin.size();
    }
    lz_hsp* res = (lz_hsp*)malloc((fin.size() ? fin.size() : 1) * sizeof(lz_hsp));
    if (!fin.empty()) memcpy(res, fin.data(), fin.size() * sizeof(lz_hsp));
    *out = res; *n_out = fin.size();
    return 0;
}
extern "C" void emul_free(void* p) { free(p); }
