// lz_ctx.hpp -- per-process device context of liblzgpu (one process per GPU).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <map>
#include <mutex>
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
    DevBuf nib;                     // 4-bit class codes, two bases per byte (byte-code scans; built when all classes are < 8)
    bool   have_nib = false;
    DevBuf two, spc;                // 2-bit codes and the 1-bit "not A,C,G,T" mask (lz_lut.hpp), rebuilt with the codes
    DevBuf two_x, spc_x;            // the same bytes in half-overlapping 64-byte blocks (k_overlap32; read by k_scan_hits for the target)
    DevBuf occ_dev;                 // [256] u32: which byte values occur
    u8     occ[256] = { 0 };        // ... on the host
    bool   has_special = false;     // some byte of the sequence is outside the 2-bit alphabet
    u32    len = 0;
    bool   have_raw = false;
    uint64_t code_key = 0;          // hash of the (class map, charToBits) the codes were built with
    uint64_t dp_key = 0;            // hash of the class map the DP codes (dp) were built with; 0 = not built / the bytes changed since
    std::vector<u8> host;           // host copy (entropy post-pass needs the raw bytes)
    u8* raw_base()  const { return raw.as<u8>()  + LZ_SEQ_PAD; }
    u8* code_base() const { return code.as<u8>() + LZ_SEQ_PAD; }
};

#define LZ_SETS 3                       // sets of the per-chunk buffers (the chunk pipeline of lzgpu_seed_hit_search)
struct LzCtx {
    bool inited = false;
    int  device = -1;
    int  num_cus = 256;                 // compute units of the device (MI355X: 256)
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr, stream3 = nullptr;   // chunk pipeline: fill + histogram (stream) | scans (stream3) | partition + phase B (stream2)
    hipStream_t dp_stream = nullptr;    // B3's own stream: lzgpu_gapped_extend(_batch) on one host thread may run beside lzgpu_seed_hit_search on another
    hipEvent_t ev_keys[LZ_SETS] = {}, ev_summ[LZ_SETS] = {}, ev_part[LZ_SETS] = {}, ev_extended[LZ_SETS] = {}, ev_init = nullptr;
    std::string last_error;

    // ---- target + position table (B1)
    SeqSlot target;
    bool have_table = false;
    lz_table_geom geom;
    LzSeedDev seed;
    DevBuf wstart, wpos;            // CSR table
    u64 num_words = 0;

    // ---- queries
    std::map<int, SeqSlot> queries; // slot -> resident query; slot -1 = B2's transient slot (a host-pointer query), LZ_TEMP_SLOT_B3 - k = problem k of a B3 batch
                                    // that came with a host pointer (map nodes are stable: a slot's address survives other slots' insertion)
#define LZ_TEMP_SLOT_B3 (-1000)
    std::mutex slots_m;             // guards look-ups / insertions in `queries` (B2 and B3 may run on two host threads)

    // ---- seed-search scratch
    DevBuf cnt, off, pk;            // per query position: raw-hit count (u32), exclusive scan (u64), packed word (u32)
    DevBuf wiv, wsk, wsv;           // position index; (block of positions | word, position) sorted by that key (seed_kernels.hip: k_pack_words)
    DevBuf blk_start;               // where each block of positions begins in the sorted list
    std::vector<u64> blk_start_host;
    u32 blk_shift = 0, blk_count = 1;
    u64* pinned = nullptr; size_t pinned_words = 0;   // host memory the device writes small results into (no staged D2H copies)
    DevBuf bins[LZ_SETS];           // the partition (high hash byte) of every hit of the chunk, written by the scan kernel (k_hist reads them)
    DevBuf keys[LZ_SETS];                 // hit keys of a chunk, discovery order (two sets of every per-chunk buffer: the chunk pipeline)
    DevBuf recs[LZ_SETS], bin_base[LZ_SETS];    // hit records partitioned by the high hash bits + the 257 partition offsets; two sets:
                                    // phase B of a chunk runs while the next chunk is filled / scanned / partitioned
    DevBuf hist[LZ_SETS], hist_part[LZ_SETS];   // per-tile partition histogram and its block sums
    DevBuf summ[LZ_SETS], scan_tasks[LZ_SETS], scan_ntasks[LZ_SETS];   // phase A: 4-byte summary per hit of the chunk; the scans that go on past their first window
    DevBuf lut;                     // phase-A tables (lz_lut.hpp)
    DevBuf sort_tmp, scan_tmp;
    DevBuf diag_end;                // [LZ_DIAG_SIZE]
    DevBuf score_tab;               // [32*32] s32
    DevBuf win_tab;                 // the same for lzgpu_window_search (its own: the two callers may use different matrices)
    DevBuf hsp_out, hsp_count;      // candidates + counter
    DevBuf hsp_mc;                  // [n][5]: A/C/G/T match counts of the candidates (entropy inputs) + probe index
    DevBuf dev_counters;            // u64[8]
    DevBuf tb_keys, tb_vals, tb_keys2, tb_vals2;   // table build scratch
    u32 n_owners = 1, owner = 0;      // bucket ownership (lzgpu_set_bucket_owner)
    std::vector<u64> last_order;      // two sort words per HSP of the last search
    int min_scan_mode = 0;            // lzgpu_set_scan_mode
    int last_scan_mode = -1;          // phase-A scan mode of the last search (0/1: look-up tables without/with special masks, 2: byte codes)
    u64 hit_capacity = (1ull << 31);    // hits per chunk (LZGPU_HIT_CAPACITY): 2^28 -> 2^30 took 10 ms off the 50 Mbp step (fewer launches, fewer passes over the sorted words), 2^30 -> 2^31 another 5.5 (a 50 Mbp strand is one chunk); the buffers follow the largest chunk: 21 bytes per hit, 42 GiB at most
    u64 hsp_capacity = (1ull << 24);

    lz_counters counters = {};      // B2 and B3 write disjoint fields (possibly from two host threads); lzgpu_counters copies under counters_m
    std::mutex counters_m;
    KernelTimer timer;              // B1 / B2 launches (the caller's thread)
    KernelTimer dp_timer;           // B3 launches (dp_stream; possibly another host thread)
};

LzCtx& lz_ctx();
int lz_bind_thread();               // brings the context up if need be and binds the CALLING thread to its device (every entry point)
int lz_fail(int code, const char* fmt, ...);
#define LZ_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) \
    return lz_fail(LZGPU_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); } while (0)

// ---- launchers implemented in seed_kernels.hip (all asynchronous on ctx.stream) ----
int lzk_encode(LzCtx& c, const u8* raw, u8* code, u32 len, const u8* cls256_dev, hipStream_t st = nullptr, KernelTimer* timer = nullptr);   // defaults: c.stream, c.timer
int lzk_pack_nibbles(LzCtx& c, const u8* code_alloc, u8* nib, size_t nbytes);
int lzk_table_build(LzCtx& c);
int lzk_table_export(LzCtx& c, u32* last_dev, u32* prev_dev, u32 prev_entries);
int lzk_count_hits(LzCtx& c, const u8* qcode, u32 lo, u32 hi, u32* cnt, u32* pk, u32* iv, u32* sk, u32* sv, u64* valid_words_dev);   // honours c.n_owners / c.owner
int lzk_scan_counts(LzCtx& c, const u32* cnt, u64* off, u32 n);
int lzk_sample_offsets(LzCtx& c, const u64* off, const u32* cnt, u32 n, u32 stride, u32 ns, u64* out);
int lzk_fill_hits(LzCtx& c, u32 lo, u32 i0, u32 i1, const u32* sk, const u32* sv, u32 n, const u64* off, u64 base, u64* keys, hipStream_t st);
int lzk_hsp_match_counts(LzCtx& c, const LzHspRec* recs, const u32* n_rec_dev, u32 cap, u32 launch_for,
                         const u8* traw, const u8* qraw, const u8* tcode, const u8* qcode, u32* counts, hipStream_t s);
struct LzLutParams; struct LzLutEntry;
#ifndef LZ_PP_TILE_HOST
#define LZ_PP_TILE_HOST 16384       // hits per tile of k_hist / k_partition (sizes the partition histogram); 8192 with 512 lanes: 24.5 ms per step, 16384 with 1024: 21.7
#endif
int lzk_pack2(LzCtx& c, const u8* code_base, const u8* raw_base, u32 len, u8* two, u8* spc, u32 nmask, u32* flags256);
int lzk_overlap32(LzCtx& c, const u8* src, u8* dst, size_t nblocks);
int lzk_hist(LzCtx& c, const u8* bins, u64 n, u32* hist, u32* part, u32* bin_base, hipStream_t st);
int lzk_scan_reserve(LzCtx& c, int set, int mode, u64 max_n);
int lzk_scan_hits(LzCtx& c, int set, int mode, const LzExtendParams& P, const LzLutParams& Q, const u64* keys, u64 n,
                  const s32* score_tab, const LzLutEntry* lut, u8* bins, hipStream_t st);     // -> c.summ[set], and the partition byte of every hit -> bins
int lzk_partition(LzCtx& c, int set, const u64* keys, u64 n, const u32* hist, const u32* part, u64* recs, hipStream_t st);
int lzk_settle(LzCtx& c, const LzExtendParams& P, const u64* recs, const u32* bin_base, u32* diag_end,
               const s32* score_tab, LzHspRec* out, u32* out_count, u32 out_cap, u64* counters, hipStream_t s);
