// lz_host.hpp -- the pure-host pieces of the seed stage (no HIP calls): seed compilation, score
// class compression, chunk planning and the per-HSP finish.  Kept free of device code so that
// tests/emul can link them and check them on a machine without a GPU.
#pragma once
#include <vector>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <atomic>
#include <functional>
#include <system_error>
#include <vector>
#include "lz_common.hpp"
#include "../../include/lzgpu.h"

int  lzh_seed_to_dev(const lz_seed_desc* sd, LzSeedDev& d);
void lzh_make_cls(const u8* score_class /*[256] or NULL*/, const int8_t ctb[256], u8 cls[256]);
int  lzh_score_classes(const s32* sub, u8 rowc[256], u8 colc[256], s32 tab[LZ_NCLASS * LZ_NCLASS]);
u32  lzh_small_classes(const u8 rowc[256], const u8 colc[256]);    // 1 when every class id is < 8
double lzh_hsp_entropy(const u8* s, const u8* t, int len);

// ---- phase-A look-up tables (lz_lut.hpp).  tocc / qocc say which byte values occur in the target / query.
// lzh_lut_eligible: 1 when the scans of this (matrix, xDrop, sequences) can run four bases per step on 2-bit
// codes -- M4 receives the 4 x 4 matrix over charToBits codes -- else 0 (the byte-code scans run instead).
struct LzLutEntry;
int  lzh_lut_eligible(const s32* sub, const int8_t ctb[256], const u8 tocc[256], const u8 qocc[256], s32 xdrop, s32 M4[16]);
void lzh_lut_build(const s32 M4[16], s32 xdrop, LzLutEntry* tab /*[LZ_LUT_TOTAL]*/);

struct LzChunk { u32 i0, i1; u64 base, nh; };
// Split query positions [0,n) into chunks of at most cap raw hits.  off_at(i) must return the
// exclusive prefix sum of the per-position hit counts at i (i in [0,n]); samples are taken every
// S positions first and refined only where one S-block alone exceeds cap.
template <class OffAt>
int lzh_plan_chunks(u32 n, u64 cap, u32 S, OffAt&& off_at, std::vector<LzChunk>& chunks)
{
    chunks.clear();
    const u32 ns = (n + S - 1) / S;
    auto samp = [&](u32 s) -> u64 { u64 i = (u64)s * S; return off_at(i < n ? (u32)i : n); };
    u32 s0 = 0;
    while (s0 < ns) {
        u32 s1 = s0;
        while (s1 < ns && samp(s1 + 1) - samp(s0) <= cap) s1++;
        if (s1 > s0) {
            u32 i0 = s0 * S, i1 = ((u64)s1 * S < n) ? s1 * S : n;
            if (samp(s1) > samp(s0)) chunks.push_back({ i0, i1, samp(s0), samp(s1) - samp(s0) });
            s0 = s1;
            continue;
        }
        u32 b0 = s0 * S, b1 = ((u64)(s0 + 1) * S < n) ? (s0 + 1) * S : n;
        u32 p0 = b0;
        while (p0 < b1) {
            u32 p1 = p0;
            while (p1 < b1 && off_at(p1 + 1) - off_at(p0) <= cap) p1++;
            if (p1 == p0) return LZGPU_NH_HITS_OVERFLOW;
            if (off_at(p1) > off_at(p0)) chunks.push_back({ p0, p1, off_at(p0), off_at(p1) - off_at(p0) });
            p0 = p1;
        }
        s0++;
    }
    return 0;
}

// Candidate HSPs (any order) -> what the reference's reporter sees, in discovery order:
// (query position of the seed hit ascending, probe index, target position descending), entropy
// adjustment (src/seed_search.c:2851-2874) and the score threshold (:2907-2933).
// match_counts (optional, [n_rec][5]): per candidate the number of positions where target and query
// carry the same byte 'A','C','G','T' -- the only thing the entropy needs from the sequences
// (src/dna_utilities.c:2899-2912) -- and the index of the probe that produced the seed hit; when NULL
// both are derived here from thost/qhost.
int lzh_finish_hsps(const LzHspRec* recs, u32 n_rec, const u8* thost, const u8* qhost,
                    const LzSeedDev& sd, const int8_t ctb[256], s32 K, int entropic,
                    std::vector<lz_hsp>& out, const u32* match_counts, std::vector<u64>* order_out = nullptr);
double lzh_entropy_from_counts(int cA, int cC, int cG, int cT, int len);

// std::sort on four threads (quarters, then two merges) -- eight from 128 K elements on (a 200 Mbp strand has 300 K anchors: 6.8 -> 3.9 ms) --
// for the host phases during which the GPU waits; `less` must be a strict weak order (elements it ties are interchangeable for the callers).
#include <algorithm>
#include <thread>
template <class It, class Less>
void lzh_sort4(It first, It last, Less less)
{
    const size_t n = (size_t)(last - first);
    if (n < 16384) { std::sort(first, last, less); return; }
    const int parts = n >= 131072 ? 8 : 4;
    It q[9];
    for (int k = 0; k <= parts; k++) q[k] = first + (n * (size_t)k) / (size_t)parts;
    {
        std::thread th[7];
        for (int k = 1; k < parts; k++) th[k - 1] = std::thread([&, k] { std::sort(q[k], q[k + 1], less); });
        std::sort(q[0], q[1], less);
        for (int k = 1; k < parts; k++) th[k - 1].join();
    }
    for (int w = 1; w < parts; w *= 2) {                       // runs of w parts -> runs of 2w, the merges of a level side by side
        std::thread th[3]; int nt = 0;
        for (int k = 2 * w; k + w <= parts; k += 2 * w) th[nt++] = std::thread([&, k, w] { std::inplace_merge(q[k], q[k + w], q[k + 2 * w], less); });
        std::inplace_merge(q[0], q[w], q[2 * w], less);
        for (int k = 0; k < nt; k++) th[k].join();
    }
}

// A few helper threads for the host loops of one big problem (a 50 Mbp strand: 79 k anchors, 1150 alignments built per
// round).  They sleep on a condition variable between the rounds and spin during a commit pass (begin_burst /
// end_burst): a fork-join then costs a few microseconds, so that loops of a few hundred items are worth handing out
// (starting a std::thread per loop cost 30-50 us each: round 3's attempts at this lost what they won).
class ForkJoin {
    std::vector<std::thread> th;
    std::mutex m; std::condition_variable cv;
    std::atomic<bool> burst{false}, quit{false};
    std::atomic<unsigned long long> gen{0};
    std::atomic<size_t> next{0}; std::atomic<int> busy{0};
    size_t n = 0, chunk = 1; const std::function<void(size_t, size_t)>* fn = nullptr;
    void drain() { for (;;) { const size_t a = next.fetch_add(chunk); if (a >= n) break; (*fn)(a, a + chunk < n ? a + chunk : n); } }
    void worker()
    {
        unsigned long long seen = 0;
        for (;;) {
            if (!burst.load(std::memory_order_acquire)) {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return burst.load() || quit.load(); });
            }
            if (quit.load()) return;
            const unsigned long long g = gen.load(std::memory_order_acquire);
            if (g == seen) { std::this_thread::yield(); continue; }
            seen = g;
            drain();
            busy.fetch_sub(1, std::memory_order_acq_rel);
        }
    }
public:
    explicit ForkJoin(unsigned workers) { try { for (unsigned k = 0; k < workers; k++) th.emplace_back([this] { worker(); }); } catch (const std::system_error&) {} }
    ~ForkJoin() { { std::lock_guard<std::mutex> lk(m); quit.store(true); } cv.notify_all(); for (auto& t : th) t.join(); }
    unsigned threads() const { return (unsigned)th.size(); }
    void begin_burst() { if (th.empty()) return; { std::lock_guard<std::mutex> lk(m); burst.store(true, std::memory_order_release); } cv.notify_all(); }
    void end_burst() { burst.store(false, std::memory_order_release); }
    // f(lo, hi) over [0, count) in chunks, on the helpers and the caller; returns when every chunk is done
    void run(size_t count, size_t chunk_, const std::function<void(size_t, size_t)>& f)
    {
        if (th.empty() || !burst.load() || count <= chunk_) { f(0, count); return; }
        n = count; chunk = chunk_; fn = &f; next.store(0);
        busy.store((int)th.size());
        gen.fetch_add(1, std::memory_order_acq_rel);
        drain();
        while (busy.load(std::memory_order_acquire) > 0) std::this_thread::yield();
    }
};
