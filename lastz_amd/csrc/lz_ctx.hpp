// lz_ctx.hpp -- per-process device context of liblzgpu (one process per GPU).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <map>
#include "lz_common.hpp"
#include "../../include/lzgpu.h"

struct DevBuf {                     // growable device allocation
    void*  p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes);       // 0 or LZGPU_ERR_OOM; contents are NOT preserved on growth
    void release();
    template <class T> T* as() const { return (T*)p; }
};

struct KernelTimer {
    struct Pending { int id; hipEvent_t a, b; hipStream_t s; };
    bool enabled = false;
    std::vector<std::string> names;
    std::vector<uint64_t> launches;
    std::vector<double> ms;
    std::vector<Pending> pending;
    std::map<hipStream_t, std::vector<hipEvent_t>> pool;   // an event is only ever recorded on one stream
    int  id_of(const char* name);
    void begin(const char* name, hipStream_t s);
    void end(hipStream_t s);
    void resolve();                 // after a stream sync: fold pending events into totals
    void reset();
    hipEvent_t get_event(hipStream_t s);
    int cur = -1; hipEvent_t cur_a = nullptr;
};

struct SeqSlot {                    // a sequence resident in HBM
    DevBuf raw;                     // LZ_SEQ_PAD + len + LZ_SEQ_PAD bytes
    DevBuf code;                    // same geometry, code bytes (see lz_common.hpp)
    DevBuf dp;                      // same geometry, DP score-class codes (unmasked scoring), B3 only
    DevBuf nib;                     // 4-bit class codes, two bases per byte (phase A; built when all classes are < 8)
    bool   have_nib = false;
    u32    len = 0;
    bool   have_raw = false;
    uint64_t code_key = 0;          // hash of the (class map, charToBits) the codes were built with
    std::vector<u8> host;           // host copy (entropy post-pass needs the raw bytes)
    u8* raw_base()  const { return raw.as<u8>()  + LZ_SEQ_PAD; }
    u8* code_base() const { return code.as<u8>() + LZ_SEQ_PAD; }
};

struct LzCtx {
    bool inited = false;
    int  device = -1;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // phase B of chunk c overlaps fill/probe/sort of chunk c+1
    hipEvent_t ev_sorted[2] = { nullptr, nullptr }, ev_extended[2] = { nullptr, nullptr }, ev_init = nullptr;
    std::string last_error;

    // ---- target + position table (B1)
    SeqSlot target;
    bool have_table = false;
    lz_table_geom geom;
    LzSeedDev seed;
    DevBuf wstart, wpos;            // CSR table
    u64 num_words = 0;

    // ---- queries
    std::map<int, SeqSlot> queries; // slot -> resident query; slot -1 = transient

    // ---- seed-search scratch
    DevBuf cnt, off, pk;            // per query position: raw-hit count (u32), exclusive scan (u64), packed word (u32)
    DevBuf wiv, wsk, wsv;           // position index; (word, position) sorted by word
    u64* pinned = nullptr; size_t pinned_words = 0;   // host memory the device writes small results into (no staged D2H copies)
    DevBuf keys_a, keys_b;          // hit keys, double buffer for the radix sort
    DevBuf summ_a, summ_b;          // phase-A summaries, travelling with the keys
    DevBuf keys_b2, summ_b2;        // second output set (double buffering across chunks)
    DevBuf sort_tmp, scan_tmp;
    DevBuf diag_end;                // [LZ_DIAG_SIZE]
    DevBuf score_tab;               // [32*32] s32
    DevBuf hsp_out, hsp_count;      // candidates + counter
    DevBuf hsp_mc;                  // [n][5]: A/C/G/T match counts of the candidates (entropy inputs) + probe index
    DevBuf dev_counters;            // u64[8]
    DevBuf tb_keys, tb_vals, tb_keys2, tb_vals2;   // table build scratch
    u32 n_owners = 1, owner = 0;      // bucket ownership (lzgpu_set_bucket_owner)
    std::vector<u64> last_order;      // two sort words per HSP of the last search
    u64 hit_capacity = (1ull << 28);
    u64 hsp_capacity = (1ull << 24);

    lz_counters counters = {};
    KernelTimer timer;
};

LzCtx& lz_ctx();
int lz_fail(int code, const char* fmt, ...);
#define LZ_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) \
    return lz_fail(LZGPU_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); } while (0)

// ---- launchers implemented in seed_kernels.hip (all asynchronous on ctx.stream) ----
int lzk_encode(LzCtx& c, const u8* raw, u8* code, u32 len, const u8* cls256_dev);
int lzk_pack_nibbles(LzCtx& c, const u8* code_alloc, u8* nib, size_t nbytes);
int lzk_table_build(LzCtx& c);
int lzk_table_export(LzCtx& c, u32* last_dev, u32* prev_dev, u32 prev_entries);
int lzk_count_hits(LzCtx& c, const u8* qcode, u32 lo, u32 hi, u32* cnt, u32* pk, u32* iv, u32* sk, u32* sv, u64* valid_words_dev);   // honours c.n_owners / c.owner
int lzk_scan_counts(LzCtx& c, const u32* cnt, u64* off, u32 n);
int lzk_sample_offsets(LzCtx& c, const u64* off, const u32* cnt, u32 n, u32 stride, u32 ns, u64* out);
int lzk_fill_hits(LzCtx& c, u32 lo, u32 i0, u32 i1, const u32* sk, const u32* sv, u32 n, const u64* off, u64 base, u64* keys);
int lzk_hsp_match_counts(LzCtx& c, const LzHspRec* recs, const u32* n_rec_dev, u32 cap, u32 launch_for,
                         const u8* traw, const u8* qraw, const u8* tcode, const u8* qcode, u32* counts, hipStream_t s);
int lzk_probe_hits(LzCtx& c, const LzExtendParams& P, const u64* keys, u64 n, const s32* score_tab, u32* summ);
int lzk_sort_hits(LzCtx& c, u64* keys_in, u64* keys_out, u32* summ_in, u32* summ_out, u64 n);
int lzk_extend(LzCtx& c, const LzExtendParams& P, const u64* keys, const u32* summ, u32 n, u32* diag_end,
               const s32* score_tab, LzHspRec* out, u32* out_count, u32 out_cap, u64* counters, hipStream_t s);
