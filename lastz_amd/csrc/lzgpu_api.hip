// lzgpu_api.hip -- C ABI (include/lzgpu.h), device context and host-side orchestration of the
// seed stage.  The host work here is what the reference does around its hot loops and is not
// data-parallel: seed compilation (src/seeds.c), score-class compression of the 256x256 matrix,
// chunk planning, and the per-HSP finish (discovery ordering + the floating-point entropy
// adjustment of src/seed_search.c:2851-2874, src/dna_utilities.c:2888-2936).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <mutex>
#include <thread>
#include <atomic>
#include "lz_ctx.hpp"
#include "lz_host.hpp"
#include "lz_lut.hpp"

// ------------------------------------------------------------------------------ context

static LzCtx g_ctx;
static void lz_release_statics();
void lz_phase_clocks_print();                                 // seed_kernels.hip (prints only in a -DLZ_PHASE_CLOCKS build)
LzCtx& lz_ctx() { return g_ctx; }

// The last error message: one per THREAD (B3 may run on a second host thread beside B2, and the problems of a batch run on
// threads of their own), plus the newest of any thread for a caller that had no failure of its own -- lzgpu_gapped_extend_batch
// reports a worker's failure from the calling thread.  lzgpu_last_error() hands out the calling thread's own copy: no torn reads,
// never a pointer into a string another thread may reassign.
static thread_local std::string t_last_error, t_last_error_out;
static std::mutex g_err_m;
int lz_fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    t_last_error = buf;
    std::lock_guard<std::mutex> lk(g_err_m);
    g_ctx.last_error = buf;
    return code;
}

int DevBuf::ensure(size_t bytes)
{
    if (bytes <= cap && p) return 0;
    static const bool prof = getenv("LZGPU_HOSTPROF") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    const size_t old = cap;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    const auto t1 = std::chrono::steady_clock::now();
    size_t want = bytes < 256 ? 256 : bytes;
    if (want >= (64u << 20)) {                                  // big buffers in steps of 1/16 of their size: a request that
        size_t step = (size_t)1 << 22;                          // creeps up (the second strand's chunks, a few more DPs) does
        while (step * 32 <= want) step <<= 1;                   // not free and re-allocate gigabytes
        want = (want + step - 1) / step * step;
    }
    hipError_t e = hipMalloc(&p, want);
    if (prof && want >= (64u << 20)) fprintf(stderr, "[lzgpu hostprof] device buffer %zu -> %zu MiB at %p: hipFree %.1f ms, hipMalloc %.1f ms\n", old >> 20, want >> 20, p,
                                             std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
    if (e != hipSuccess) { p = nullptr; return lz_fail(LZGPU_ERR_OOM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); }
    cap = want;
    return 0;
}
void DevBuf::release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }

int KernelTimer::id_of(const char* name)
{
    for (size_t i = 0; i < names.size(); i++) if (names[i] == name) return (int)i;
    names.push_back(name); launches.push_back(0); ms.push_back(0.0);
    return (int)names.size() - 1;
}
hipEvent_t KernelTimer::get_event(hipStream_t s)
{
    auto& pl = pool[s];
    if (!pl.empty()) { hipEvent_t e = pl.back(); pl.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
void KernelTimer::begin(const char* name, hipStream_t s)
{
    if (!enabled) return;
    static const char* only = getenv("LZGPU_TIMER_ONLY");        // profiling aid: time just this kernel
    if (only && strcmp(only, name) != 0) { cur = -1; return; }
    cur = id_of(name); cur_a = get_event(s);
    (void)hipEventRecord(cur_a, s);
}
void KernelTimer::end(hipStream_t s)
{
    if (!enabled || cur < 0) return;
    hipEvent_t b = get_event(s);
    (void)hipEventRecord(b, s);
    pending.push_back({cur, cur_a, b, s});
    cur = -1; cur_a = nullptr;
}
void KernelTimer::resolve()
{
    // LZGPU_GAPS=1 (profiling aid): idle time on a stream between the end of one timed kernel and the
    // start of the next, reported when it exceeds 0.3 ms
    static const bool gaps = getenv("LZGPU_GAPS") != nullptr;
    static std::map<hipStream_t, Pending> last;
    std::vector<Pending> later;
    for (auto& p : pending) {
        float t = 0.f;
        const hipError_t e = hipEventElapsedTime(&t, p.a, p.b);
        if (e == hipErrorNotReady) { later.push_back(p); continue; }     // recorded after the last synchronisation
        if (e == hipSuccess) { ms[p.id] += t; launches[p.id]++; }
        if (gaps) {
            auto it = last.find(p.s);
            float g = 0.f;
            if (it != last.end() && hipEventElapsedTime(&g, it->second.b, p.a) == hipSuccess && g > 0.3f)
                fprintf(stderr, "[lzgpu gaps] %.2f ms idle between %s and %s\n", g, names[it->second.id].c_str(), names[p.id].c_str());
            if (it != last.end()) { pool[it->second.s].push_back(it->second.a); pool[it->second.s].push_back(it->second.b); }
            last[p.s] = p;
            continue;
        }
        pool[p.s].push_back(p.a); pool[p.s].push_back(p.b);
    }
    pending.swap(later);
}
void KernelTimer::reset() { resolve(); for (auto& x : ms) x = 0; for (auto& x : launches) x = 0; }

// HIP's current device is a property of the HOST THREAD.  The context may have been brought up on another thread
// (lzgpu_init_async's, or whichever thread called lzgpu_init first), so every entry point that touches the device
// first binds the calling thread to the library's device: without it a rank r > 0 of a multi-GPU run would allocate
// and copy on device 0 while its streams live on device r.  (hipSetDevice on the device that is already current
// is a thread-local store.)
int lz_bind_thread()
{
    if (!g_ctx.inited) { int rc = lzgpu_init(-1); if (rc) return rc; }
    LZ_HIP(hipSetDevice(g_ctx.device));
    return 0;
}
static int require_init() { return lz_bind_thread(); }

extern "C" const char* lzgpu_last_error(void)
{
    if (!t_last_error.empty()) t_last_error_out = t_last_error;
    else { std::lock_guard<std::mutex> lk(g_err_m); t_last_error_out = g_ctx.last_error; }
    return t_last_error_out.c_str();
}

extern "C" int lzgpu_probe(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return lz_fail(LZGPU_ERR_NO_DEVICE, "no HIP device visible");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return lz_fail(LZGPU_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return lz_fail(LZGPU_ERR_NO_DEVICE, "device 0 is %s, this library is built for gfx950 only", prop.gcnArchName);
    return 0;
}

static int lz_init_locked(int device_index)
{
    if (g_ctx.inited && (device_index < 0 || device_index == g_ctx.device)) return 0;
    if (g_ctx.inited) return lz_fail(LZGPU_ERR_STATE, "already bound to device %d", g_ctx.device);
    int rc = lzgpu_probe();
    if (rc) return rc;
    if (device_index < 0) {
        const char* lr = getenv("LOCAL_RANK");
        device_index = lr ? atoi(lr) : 0;
        int n = 0; (void)hipGetDeviceCount(&n);
        if (n > 0) device_index %= n;
    }
    LZ_HIP(hipSetDevice(device_index));
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, device_index) == hipSuccess && prop.multiProcessorCount > 0) g_ctx.num_cus = prop.multiProcessorCount; }
    LZ_HIP(hipStreamCreateWithFlags(&g_ctx.stream, hipStreamNonBlocking));
    LZ_HIP(hipStreamCreateWithFlags(&g_ctx.stream2, hipStreamNonBlocking));
    LZ_HIP(hipStreamCreateWithFlags(&g_ctx.stream3, hipStreamNonBlocking));
    LZ_HIP(hipStreamCreateWithFlags(&g_ctx.dp_stream, hipStreamNonBlocking));
    for (int k = 0; k < LZ_SETS; k++) {
        LZ_HIP(hipEventCreateWithFlags(&g_ctx.ev_keys[k], hipEventDisableTiming));
        LZ_HIP(hipEventCreateWithFlags(&g_ctx.ev_summ[k], hipEventDisableTiming));
        LZ_HIP(hipEventCreateWithFlags(&g_ctx.ev_part[k], hipEventDisableTiming));
        LZ_HIP(hipEventCreateWithFlags(&g_ctx.ev_extended[k], hipEventDisableTiming));
    }
    LZ_HIP(hipEventCreateWithFlags(&g_ctx.ev_init, hipEventDisableTiming));
    if (const char* sm = getenv("LZGPU_SCAN_MODE")) { const int v = atoi(sm); if (v >= 0 && v <= 2) g_ctx.min_scan_mode = v; }
    if (const char* hc = getenv("LZGPU_HIT_CAPACITY")) { const long long v = atoll(hc); if (v >= 1024 && v <= (1ll << 31)) g_ctx.hit_capacity = (u64)v; }
    g_ctx.device = device_index;
    g_ctx.inited = true;
    return 0;
}

// lzgpu_init_async: the same on a detached thread; every path into lzgpu_init first waits for it (and so does exit():
// the runtime must not be torn down under a thread that is still bringing it up)
static std::mutex g_init_mutex;
static std::atomic<int> g_async_state{0};                      // 0: never asked, 1: running, 2: finished
static void lz_wait_async() { while (g_async_state.load(std::memory_order_acquire) == 1) std::this_thread::yield(); }
extern "C" void lzgpu_init_async(int device_index)
{
    int expect = 0;
    if (!g_async_state.compare_exchange_strong(expect, 1)) return;
    atexit(lz_wait_async);
    std::thread([device_index]() {
        { std::lock_guard<std::mutex> lk(g_init_mutex); (void)lz_init_locked(device_index); }
        g_async_state.store(2, std::memory_order_release);
    }).detach();
}
extern "C" int lzgpu_init(int device_index)
{
    lz_wait_async();
    std::lock_guard<std::mutex> lk(g_init_mutex);
    const int rc = lz_init_locked(device_index);
    if (!rc) LZ_HIP(hipSetDevice(g_ctx.device));               // the caller's thread too (the context may be another thread's work)
    return rc;
}
extern "C" int lzgpu_device_index(void) { return g_ctx.inited ? g_ctx.device : -1; }

extern "C" void lzgpu_shutdown(void)
{
    LzCtx& c = g_ctx;
    if (!c.inited) return;
    (void)hipSetDevice(c.device);
    (void)hipStreamSynchronize(c.stream);
    lz_phase_clocks_print();
    if (c.stream2) (void)hipStreamSynchronize(c.stream2);
    if (c.stream3) (void)hipStreamSynchronize(c.stream3);
    if (c.dp_stream) (void)hipStreamSynchronize(c.dp_stream);
    c.timer.resolve(); c.dp_timer.resolve();
    if (c.pinned) { (void)hipHostFree(c.pinned); c.pinned = nullptr; c.pinned_words = 0; }
    DevBuf* bufs[] = { &c.target.raw, &c.target.code, &c.wstart, &c.wpos, &c.cnt, &c.off, &c.pk, &c.wiv, &c.wsk, &c.wsv, &c.lut,
                       &c.sort_tmp, &c.scan_tmp, &c.diag_end, &c.score_tab, &c.hsp_out, &c.hsp_count, &c.hsp_mc,
                       &c.dev_counters, &c.tb_keys, &c.tb_vals, &c.tb_keys2, &c.tb_vals2 };
    for (DevBuf* b : bufs) b->release();
    for (int k = 0; k < LZ_SETS; k++) { DevBuf* sb[] = { &c.bins[k], &c.keys[k], &c.recs[k], &c.bin_base[k], &c.hist[k], &c.hist_part[k], &c.summ[k], &c.scan_tasks[k], &c.scan_ntasks[k] }; for (DevBuf* b : sb) b->release(); }
    for (auto& kv : c.queries) { kv.second.raw.release(); kv.second.code.release(); kv.second.dp.release(); kv.second.nib.release(); kv.second.two.release(); kv.second.spc.release(); kv.second.occ_dev.release(); }
    c.target.dp.release(); c.target.nib.release(); c.target.two.release(); c.target.spc.release(); c.target.occ_dev.release();
    c.target.two_x.release(); c.target.spc_x.release();
    lz_release_statics();
    c.queries.clear();
    (void)hipStreamDestroy(c.stream);
    if (c.stream2) (void)hipStreamDestroy(c.stream2);
    if (c.stream3) (void)hipStreamDestroy(c.stream3);
    if (c.dp_stream) (void)hipStreamDestroy(c.dp_stream);
    c.dp_stream = nullptr;
    for (int k = 0; k < LZ_SETS; k++) {
        hipEvent_t* ev[] = { &c.ev_keys[k], &c.ev_summ[k], &c.ev_part[k], &c.ev_extended[k] };
        for (hipEvent_t* e : ev) { if (*e) (void)hipEventDestroy(*e); *e = nullptr; }
    }
    if (c.ev_init) (void)hipEventDestroy(c.ev_init);
    c.ev_init = nullptr; c.stream2 = nullptr; c.stream3 = nullptr;
    c.stream = nullptr; c.inited = false; c.have_table = false; c.device = -1;
    c.n_owners = 1; c.owner = 0; c.last_order.clear();
}

extern "C" void lzgpu_free(void* p) { free(p); }

// ------------------------------------------------------------------------------ sequences

static int slot_upload(LzCtx& c, SeqSlot& s, const u8* bytes, u32 len, bool keep_host, hipStream_t st = nullptr)
{
    int rc;
    if (!st) st = c.stream;
    size_t total = (size_t)len + 2 * LZ_SEQ_PAD + 16;
    if ((rc = s.raw.ensure(total))) return rc;
    if ((rc = s.code.ensure(total))) return rc;
    LZ_HIP(hipMemsetAsync(s.raw.p, 0, total, st));
    LZ_HIP(hipMemsetAsync(s.code.p, LZ_CODE_INVALID, total, st));
    if (len) LZ_HIP(hipMemcpyAsync(s.raw_base(), bytes, len, hipMemcpyHostToDevice, st));
    LZ_HIP(hipStreamSynchronize(st));
    s.len = len; s.have_raw = true; s.code_key = 0; s.dp_key = 0;
    if (keep_host) s.host.assign(bytes, bytes + len); else s.host.clear();
    return 0;
}

static uint64_t fnv1a(const void* p, size_t n, uint64_t h = 1469598103934665603ull)
{
    const u8* b = (const u8*)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h ? h : 1;
}

// encode slot with table cls (class | bits<<5 | invalid<<7), unless already encoded with it
static int slot_encode(LzCtx& c, SeqSlot& s, const u8 cls[256], DevBuf& cls_dev)
{
    uint64_t key = fnv1a(cls, 256);
    if (s.code_key == key) return 0;
    int rc;
    if ((rc = cls_dev.ensure(256))) return rc;
    LZ_HIP(hipMemcpyAsync(cls_dev.p, cls, 256, hipMemcpyHostToDevice, c.stream));
    LZ_HIP(hipStreamSynchronize(c.stream));          // cls is caller stack memory
    if ((rc = lzk_encode(c, s.raw_base(), s.code_base(), s.len, cls_dev.as<u8>()))) return rc;
    s.code_key = key;
    // packed 4-bit codes for phase A when the matrix has fewer than 8 classes
    bool small = true;
    for (int b = 0; b < 256; b++) if ((cls[b] & 31u) >= 8u) { small = false; break; }
    s.have_nib = false;
    if (small) {
        const size_t total = (size_t)s.len + 2 * LZ_SEQ_PAD + 16;   // the code allocation (even)
        if ((rc = s.nib.ensure(total / 2 + 64))) return rc;
        LZ_HIP(hipMemsetAsync(s.nib.p, 0, total / 2 + 64, c.stream));
        if ((rc = lzk_pack_nibbles(c, s.code.as<u8>(), s.nib.as<u8>(), total / 2))) return rc;
        s.have_nib = true;
    }
    // 2-bit codes, special mask and the set of byte values that occur (phase A on look-up tables, lz_lut.hpp)
    const u32 nmask = (u32)(((size_t)s.len + 2 * LZ_PAD2 + 7) / 8 + 16);
    if ((rc = s.two.ensure((size_t)nmask * 2 + 96))) return rc;       // (+ what k_overlap32's last block reads past the end)
    if ((rc = s.spc.ensure((size_t)nmask + 96))) return rc;
    if ((rc = s.occ_dev.ensure(256 * 4))) return rc;
    LZ_HIP(hipMemsetAsync(s.two.p, 0, (size_t)nmask * 2 + 96, c.stream));
    LZ_HIP(hipMemsetAsync(s.spc.p, 0xFF, (size_t)nmask + 96, c.stream));
    if ((rc = lzk_pack2(c, s.code_base(), s.raw_base(), s.len, s.two.as<u8>(), s.spc.as<u8>(), nmask, s.occ_dev.as<u32>()))) return rc;
    if (&s == &c.target) {                                     // (the target only: k_scan_hits reads these instead of the plain arrays)
        const size_t nb2 = ((size_t)nmask * 2 + 31) / 32, nb1 = ((size_t)nmask + 31) / 32;
        if ((rc = s.two_x.ensure(nb2 * 64 + 64))) return rc;
        if ((rc = s.spc_x.ensure(nb1 * 64 + 64))) return rc;
        if ((rc = lzk_overlap32(c, s.two.as<u8>(), s.two_x.as<u8>(), nb2))) return rc;
        if ((rc = lzk_overlap32(c, s.spc.as<u8>(), s.spc_x.as<u8>(), nb1))) return rc;
    }
    u32 flags[256];
    LZ_HIP(hipMemcpyAsync(flags, s.occ_dev.p, sizeof(flags), hipMemcpyDeviceToHost, c.stream));
    LZ_HIP(hipStreamSynchronize(c.stream));
    s.has_special = false;
    for (int b = 0; b < 256; b++) { s.occ[b] = flags[b] ? 1 : 0; if (flags[b] && (cls[b] & LZ_CODE_INVALID)) s.has_special = true; }
    return 0;
}

static DevBuf g_cls_t, g_cls_q, g_cls_tmp;
void lz_dp_release_statics();                                  // dp_kernels.hip
void lz_win_release_statics();                                // window_kernels.hip
static void lz_release_statics() { g_cls_t.release(); g_cls_q.release(); g_cls_tmp.release(); lz_dp_release_statics(); lz_win_release_statics(); g_ctx.win_tab.release(); }

// (the two helpers of B3's set-up: on B3's stream, with B3's own class-table buffer and timer)
int lz_slot_upload_public(LzCtx& c, SeqSlot& s, const u8* bytes, u32 len) { return slot_upload(c, s, bytes, len, false, c.dp_stream); }
int lz_encode_with(LzCtx& c, const u8* raw, u8* code, u32 len, const u8 cls[256])
{
    int rc = g_cls_tmp.ensure(256); if (rc) return rc;
    LZ_HIP(hipMemcpyAsync(g_cls_tmp.p, cls, 256, hipMemcpyHostToDevice, c.dp_stream));
    LZ_HIP(hipStreamSynchronize(c.dp_stream));
    if ((rc = lzk_encode(c, raw, code, len, g_cls_tmp.as<u8>(), c.dp_stream, &c.dp_timer))) return rc;
    LZ_HIP(hipStreamSynchronize(c.dp_stream));
    return 0;
}
SeqSlot* lz_query_slot(LzCtx& c, int slot, bool create)
{
    std::lock_guard<std::mutex> lk(c.slots_m);
    if (create) return &c.queries[slot];
    auto it = c.queries.find(slot);
    return it == c.queries.end() ? nullptr : &it->second;
}

extern "C" int lzgpu_query_upload(int32_t slot, const uint8_t* q, uint32_t qlen)
{
    int rc = require_init(); if (rc) return rc;
    if (slot < 0 || !q) return lz_fail(LZGPU_ERR_ARG, "bad query slot");
    if (qlen >= 0x7FFFFFFFu) return LZGPU_NH_SIZE;
    return slot_upload(g_ctx, *lz_query_slot(g_ctx, slot, true), q, qlen, true);
}

// ------------------------------------------------------------------------------ B1

extern "C" int lzgpu_table_prepare(const uint8_t* t, uint32_t tlen, uint32_t start, uint32_t end,
                                   const int8_t char_to_bits[256], const lz_seed_desc* seed, uint32_t step)
{
    int rc = require_init(); if (rc) return rc;
    LzCtx& c = g_ctx;
    if (!t || !seed || !char_to_bits) return lz_fail(LZGPU_ERR_ARG, "null argument");
    if (step < 1) return lz_fail(LZGPU_ERR_ARG, "in build_seed_position_table(), step can't be %u", step);
    if (end == 0) end = tlen;
    if (end <= start) return lz_fail(LZGPU_ERR_ARG, "interval is void (%u..%u)", start, end);
    if (end > tlen)   return lz_fail(LZGPU_ERR_ARG, "interval end is bad (%u>%u)", end, tlen);
    if (tlen >= 0x7FFFFFFFu) return LZGPU_NH_SIZE;
    if ((rc = lzh_seed_to_dev(seed, c.seed))) return rc;

    c.have_table = false;
    memset(&c.geom, 0, sizeof(c.geom));
    c.geom.tlen = tlen; c.geom.start = start; c.geom.end = end; c.geom.step = step; c.geom.seed = *seed;
    memcpy(c.geom.char_to_bits, char_to_bits, 256);
    if ((rc = slot_upload(c, c.target, t, tlen, true))) return rc;
    u8 cls[256]; lzh_make_cls(nullptr, char_to_bits, cls);
    if ((rc = slot_encode(c, c.target, cls, g_cls_t))) return rc;
    if ((rc = lzk_table_build(c))) return rc;
    c.geom.num_words = c.num_words;
    c.have_table = true;
    return 0;
}

extern "C" int lzgpu_target_upload(const uint8_t* t, uint32_t tlen)
{
    int rc = require_init(); if (rc) return rc;
    LzCtx& c = g_ctx;
    if (!t) return lz_fail(LZGPU_ERR_ARG, "null argument");
    if (tlen >= 0x7FFFFFFFu) return LZGPU_NH_SIZE;
    c.have_table = false;
    memset(&c.geom, 0, sizeof(c.geom));
    c.geom.tlen = tlen;
    return slot_upload(c, c.target, t, tlen, true);
}

extern "C" int lzgpu_table_rebuild(void)
{
    LzCtx& c = g_ctx;
    if (!c.have_table) return lz_fail(LZGPU_ERR_STATE, "no position table");
    int rc = lz_bind_thread(); if (rc) return rc;
    rc = lzk_table_build(c);
    if (!rc) c.geom.num_words = c.num_words;
    return rc;
}

extern "C" uint64_t lzgpu_table_num_words(void) { return g_ctx.have_table ? g_ctx.num_words : 0; }

extern "C" int lzgpu_table_export(uint32_t* last, uint32_t* prev)
{
    LzCtx& c = g_ctx;
    if (!c.have_table) return lz_fail(LZGPU_ERR_STATE, "no position table");
    u32 nwords = 1u << c.seed.weight;
    u32 adj = c.geom.start - (c.geom.start % c.geom.step);
    u32 prev_entries = 1 + (c.geom.end - adj) / c.geom.step;       // src/pos_table.c:1065
    DevBuf dl, dp; int rc;
    if ((rc = lz_bind_thread())) return rc;
    if (last && (rc = dl.ensure((size_t)nwords * 4))) return rc;
    if (prev && (rc = dp.ensure((size_t)prev_entries * 4))) { dl.release(); return rc; }
    rc = lzk_table_export(c, last ? dl.as<u32>() : nullptr, prev ? dp.as<u32>() : nullptr, prev_entries);
    if (!rc) {
        if (last) (void)hipMemcpyAsync(last, dl.p, (size_t)nwords * 4, hipMemcpyDeviceToHost, c.stream);
        if (prev) (void)hipMemcpyAsync(prev, dp.p, (size_t)prev_entries * 4, hipMemcpyDeviceToHost, c.stream);
        if (hipStreamSynchronize(c.stream) != hipSuccess) rc = lz_fail(LZGPU_ERR_HIP, "table export copy failed");
    }
    dl.release(); dp.release();
    return rc;
}

extern "C" int lzgpu_table_geom(lz_table_geom* out)
{
    if (!g_ctx.have_table || !out) return lz_fail(LZGPU_ERR_STATE, "no position table");
    *out = g_ctx.geom; return 0;
}

extern "C" int lzgpu_table_adopt(const lz_table_geom* g)
{
    int rc = require_init(); if (rc) return rc;
    LzCtx& c = g_ctx;
    if (!g) return LZGPU_ERR_ARG;
    if ((rc = lzh_seed_to_dev(&g->seed, c.seed))) return rc;
    c.have_table = false; c.geom = *g; c.num_words = g->num_words;
    size_t total = (size_t)g->tlen + 2 * LZ_SEQ_PAD + 16;
    if ((rc = c.target.raw.ensure(total))) return rc;
    if ((rc = c.target.code.ensure(total))) return rc;
    LZ_HIP(hipMemsetAsync(c.target.raw.p, 0, total, c.stream));
    LZ_HIP(hipMemsetAsync(c.target.code.p, LZ_CODE_INVALID, total, c.stream));
    c.target.len = g->tlen; c.target.have_raw = true; c.target.code_key = 0; c.target.dp_key = 0; c.target.host.clear();
    if ((rc = c.wstart.ensure(((size_t)(1u << c.seed.weight) + 1) * 4))) return rc;
    if ((rc = c.wpos.ensure((size_t)(g->num_words ? g->num_words : 1) * 4))) return rc;
    LZ_HIP(hipStreamSynchronize(c.stream));
    return 0;
}

extern "C" int lzgpu_table_buffers(void* dev_ptr[3], uint64_t bytes[3])
{
    LzCtx& c = g_ctx;
    if (!c.target.have_raw || !c.wstart.p) return lz_fail(LZGPU_ERR_STATE, "no table buffers");
    dev_ptr[0] = c.target.raw_base(); bytes[0] = c.geom.tlen;
    dev_ptr[1] = c.wstart.p;          bytes[1] = ((uint64_t)(1u << c.seed.weight) + 1) * 4;
    dev_ptr[2] = c.wpos.p;            bytes[2] = (uint64_t)c.num_words * 4;
    return 0;
}

extern "C" int lzgpu_table_commit(void)
{
    LzCtx& c = g_ctx;
    if (!c.target.have_raw || !c.wstart.p) return lz_fail(LZGPU_ERR_STATE, "no table buffers");
    { int rc = lz_bind_thread(); if (rc) return rc; }
    // the entropy finish needs the target bytes on the host as well (the target itself does not
    // change between commits of the same geometry: copied back once)
    if (c.target.host.size() != c.geom.tlen) {
        c.target.host.resize(c.geom.tlen);
        if (c.geom.tlen) LZ_HIP(hipMemcpy(c.target.host.data(), c.target.raw_base(), c.geom.tlen, hipMemcpyDeviceToHost));
        c.target.code_key = 0; c.target.dp_key = 0;
    }
    c.have_table = true;
    return 0;
}

extern "C" int lzgpu_device_copy(void* dst, const void* src, uint64_t bytes)
{
    int rc = require_init(); if (rc) return rc;
    if (!bytes) return 0;
    LZ_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, g_ctx.stream));
    LZ_HIP(hipStreamSynchronize(g_ctx.stream));
    return 0;
}

// ------------------------------------------------------------------------------ B2

#include <chrono>
struct HostProf {
    bool on; double t[16]; const char* names[16] = { nullptr }; int n = 0;
    std::chrono::steady_clock::time_point last;
    HostProf() { on = getenv("LZGPU_HOSTPROF") != nullptr; for (auto& x : t) x = 0; }
    void start() { if (on) last = std::chrono::steady_clock::now(); }
    void lap(int k, const char* name) {
        if (!on) return;
        auto now = std::chrono::steady_clock::now();
        t[k] += std::chrono::duration<double, std::milli>(now - last).count(); names[k] = name; if (k >= n) n = k + 1; last = now;
    }
    ~HostProf() { if (on) for (int k = 0; k < n; k++) if (names[k]) fprintf(stderr, "[lzgpu hostprof] %-28s %10.2f ms total\n", names[k], t[k]); }
};
static HostProf g_hp;

extern "C" int lzgpu_seed_hit_search(const lz_search_args* a, lz_hsp** out, uint64_t* n_out)
{
    int rc = require_init(); if (rc) return rc;
    LzCtx& c = g_ctx;
    if (!a || !out || !n_out || !a->sub) return lz_fail(LZGPU_ERR_ARG, "null argument");
    *out = nullptr; *n_out = 0;
    if (!c.have_table) return lz_fail(LZGPU_ERR_STATE, "lzgpu_table_prepare has not been called");

    g_hp.start();
    // ---- query
    SeqSlot* qs;
    if (a->query) {
        if (a->qlen >= 0x7FFFFFFFu) return LZGPU_NH_SIZE;
        qs = lz_query_slot(c, -1, true);
        if ((rc = slot_upload(c, *qs, a->query, a->qlen, false))) return rc;
    } else {
        qs = a->query_slot < 0 ? nullptr : lz_query_slot(c, a->query_slot, false);
        if (!qs) return lz_fail(LZGPU_ERR_ARG, "query slot %d is empty", a->query_slot);
    }
    const u8* qhost = a->query ? a->query : qs->host.data();
    const u32 qlen = qs->len;
    u32 lo = a->start, hi = a->end ? a->end : qlen;
    if (hi <= lo) return lz_fail(LZGPU_ERR_ARG, "in seed_hit_search(), interval is void (%u-%u)", lo, hi);
    if (hi > qlen) return lz_fail(LZGPU_ERR_ARG, "in seed_hit_search(), interval end is bad (%u>%u)", hi, qlen);

    g_hp.lap(7, "query slot");
    // ---- scoring classes, codes
    // (the class compression and the 4 KiB table upload are skipped when the caller passes the same
    // matrix as last time -- lastz passes maskedScoring for every strand of every query)
    static std::vector<s32> last_sub; static u8 rowc[256], colc[256]; static s32 tab[LZ_NCLASS * LZ_NCLASS];
    u8 cls[256];
    if (last_sub.size() != 65536 || memcmp(last_sub.data(), a->sub, 65536 * sizeof(s32)) != 0 || !c.score_tab.p) {
        last_sub.clear();
        if ((rc = lzh_score_classes(a->sub, rowc, colc, tab))) return rc;
        if ((rc = c.score_tab.ensure(sizeof(tab)))) return rc;
        LZ_HIP(hipMemcpyAsync(c.score_tab.p, tab, sizeof(tab), hipMemcpyHostToDevice, c.stream));
        LZ_HIP(hipStreamSynchronize(c.stream));
        last_sub.assign(a->sub, a->sub + 65536);
    }
    g_hp.lap(8, "score classes + table upload");
    lzh_make_cls(rowc, c.geom.char_to_bits, cls);
    if ((rc = slot_encode(c, c.target, cls, g_cls_t))) return rc;
    lzh_make_cls(colc, c.geom.char_to_bits, cls);
    if ((rc = slot_encode(c, *qs, cls, g_cls_q))) return rc;

    g_hp.lap(0, "upload+classes+encode");
    const u32 L = (u32)c.seed.length;
    if (qlen < L) return 0;                                     // src/seed_search.c:486-487

    // ---- state
    const u32 n = hi - lo;
    if ((rc = c.cnt.ensure((size_t)n * 4))) return rc;
    if ((rc = c.off.ensure((size_t)n * 8))) return rc;
    if ((rc = c.pk.ensure((size_t)n * 4))) return rc;
    if ((rc = c.wiv.ensure((size_t)n * 4))) return rc;
    if ((rc = c.wsk.ensure((size_t)n * 4))) return rc;
    if ((rc = c.wsv.ensure((size_t)n * 4))) return rc;
    if ((rc = c.diag_end.ensure((size_t)LZ_DIAG_SIZE * 4))) return rc;
    if ((rc = c.dev_counters.ensure(8 * 8))) return rc;
    if ((rc = c.hsp_count.ensure(4))) return rc;
    LZ_HIP(hipMemsetAsync(c.diag_end.p, 0, (size_t)LZ_DIAG_SIZE * 4, c.stream));      // empty_diag_hash
    LZ_HIP(hipMemsetAsync(c.dev_counters.p, 0, 64, c.stream));
    LZ_HIP(hipMemsetAsync(c.hsp_count.p, 0, 4, c.stream));
    u64* d_counters = c.dev_counters.as<u64>();                 // [0]=extensions [1]=bp [2]=words

    // ---- 1. count + scan
    if ((rc = lzk_count_hits(c, qs->code_base(), lo, hi, c.cnt.as<u32>(), c.pk.as<u32>(), c.wiv.as<u32>(), c.wsk.as<u32>(), c.wsv.as<u32>(), d_counters + 2))) return rc;
    if ((rc = lzk_scan_counts(c, c.cnt.as<u32>(), c.off.as<u64>(), n))) return rc;

    // total and the sampled prefix sums for the chunk plan come back through pinned memory the device
    // writes itself: one synchronisation, no staged device-to-host copies
    const u32 S = 4096;
    const u32 ns = (n + S - 1) / S;
    if (c.pinned_words < (size_t)ns + 16) {
        if (c.pinned) (void)hipHostFree(c.pinned);
        c.pinned = nullptr; c.pinned_words = 0;
        LZ_HIP(hipHostMalloc((void**)&c.pinned, ((size_t)ns + 16) * 8, hipHostMallocDefault));
        c.pinned_words = (size_t)ns + 16;
    }
    if ((rc = lzk_sample_offsets(c, c.off.as<u64>(), c.cnt.as<u32>(), n, S, ns, c.pinned))) return rc;
    c.blk_start_host.assign((size_t)c.blk_count + 1, 0);       // (a chunk's launch of the fill kernel covers its blocks' range of the sorted list)
    LZ_HIP(hipMemcpyAsync(c.blk_start_host.data(), c.blk_start.p, ((size_t)c.blk_count + 1) * 8, hipMemcpyDeviceToHost, c.stream));
    LZ_HIP(hipStreamSynchronize(c.stream));
    g_hp.lap(1, "count+scan (sync)");
    c.timer.resolve();
    const u64 total_hits = c.pinned[0];
    const u64* samp = c.pinned + 1;

    // ---- 2. chunk plan: [i0,i1) in query positions with at most hit_capacity hits each.
    // Prefix sums are sampled every S positions; finer values are fetched only if a single S-block
    // exceeds the capacity.
    std::vector<LzChunk> chunks;
    hipError_t fetch_err = hipSuccess;
    auto off_at = [&](u32 i) -> u64 {
        if (i >= n) return total_hits;
        if (i % S == 0) return samp[i / S];
        u64 v = 0;
        hipError_t e = hipMemcpy(&v, c.off.as<u64>() + i, 8, hipMemcpyDeviceToHost);
        if (e != hipSuccess) fetch_err = e;
        return v;
    };
    if ((rc = lzh_plan_chunks(n, c.hit_capacity, S, off_at, chunks))) return rc;
    if (fetch_err != hipSuccess) return lz_fail(LZGPU_ERR_HIP, "prefix fetch failed: %s", hipGetErrorString(fetch_err));

    g_hp.lap(2, "chunk plan");
    u64 max_chunk = 0;
    for (auto& ch : chunks) if (ch.nh > max_chunk) max_chunk = ch.nh;
    // LZGPU_OVERLAP=1: the chunk pipeline over three streams (below); default: one stream, one buffer set
    static const bool overlap = getenv("LZGPU_OVERLAP") != nullptr && getenv("LZGPU_SERIAL") == nullptr;
    const int nsets = overlap ? (int)std::min<size_t>(chunks.size() ? chunks.size() : 1, LZ_SETS) : 1;
    if (max_chunk) {
        if ((rc = c.keys[0].ensure((size_t)max_chunk * 8))) return rc;
        if ((rc = c.bins[0].ensure((size_t)max_chunk + 64))) return rc;      // k_fill_hits writes a partition byte per hit on every path (the plain-hit path included)
        if (a->extend) {
            const size_t ntiles = (size_t)((max_chunk + LZ_PP_TILE_HOST - 1) / LZ_PP_TILE_HOST), nblocks = (ntiles + 255) / 256;
            for (int k = 0; k < nsets; k++) {
                if ((rc = c.keys[k].ensure((size_t)max_chunk * 8))) return rc;
                if ((rc = c.bins[k].ensure((size_t)max_chunk + 64))) return rc;
                if ((rc = c.recs[k].ensure((size_t)max_chunk * 8))) return rc;
                if ((rc = c.bin_base[k].ensure(257 * 4))) return rc;
                if ((rc = c.hist[k].ensure(ntiles * 256 * 4))) return rc;
                if ((rc = c.hist_part[k].ensure(nblocks * 256 * 4))) return rc;
            }
        }
    }
    const u32 out_cap = (u32)std::min<u64>(c.hsp_capacity, 0xFFFFFFF0ull);
    if (a->extend && (rc = c.hsp_out.ensure((size_t)out_cap * sizeof(LzHspRec)))) return rc;

    LzExtendParams P;
    P.tcode = c.target.code_base(); P.tlen = c.geom.tlen;
    P.qcode = qs->code_base();      P.qlen = qlen;
    P.xdrop = a->xdrop; P.min_score = a->hsp_threshold; P.seed_len = L;
    P.cls8 = lzh_small_classes(rowc, colc);
    const bool nibs = P.cls8 && c.target.have_nib && qs->have_nib;
    P.tnib = nibs ? c.target.nib.as<u8>() : nullptr; P.qnib = nibs ? qs->nib.as<u8>() : nullptr;

    // phase A: four bases per step on 2-bit codes when the matrix, xDrop and the bytes that occur allow it
    // (mode 0 / 1 = without / with special-byte masks), else the byte-code scans (mode 2)
    LzLutParams Q;
    Q.t2 = c.target.two.as<u8>(); Q.q2 = qs->two.as<u8>(); Q.tsp = c.target.spc.as<u8>(); Q.qsp = qs->spc.as<u8>(); Q.xdrop = a->xdrop;
    Q.t2x = c.target.two_x.as<u8>(); Q.tspx = c.target.spc_x.as<u8>();
    Q.tcode = P.tcode; Q.qcode = P.qcode;
    int mode = 2;
    if (a->extend) {
        s32 M4[16];
        if (lzh_lut_eligible(a->sub, c.geom.char_to_bits, c.target.occ, qs->occ, a->xdrop, M4))
            mode = (c.target.has_special || qs->has_special) ? 1 : 0;
        if (mode < c.min_scan_mode) mode = c.min_scan_mode;         // lzgpu_set_scan_mode (tests): 1 = masks even without specials, 2 = byte-code scans
        if (mode < 2) {
            static std::vector<LzLutEntry> lut_host; static s32 lut_m4[16]; static s32 lut_x = -1;
            if (lut_x != a->xdrop || memcmp(lut_m4, M4, sizeof(M4)) != 0 || !c.lut.p || lut_host.empty()) {
                lut_host.resize(LZ_LUT_TOTAL);
                lzh_lut_build(M4, a->xdrop, lut_host.data());
                if ((rc = c.lut.ensure(lut_host.size() * sizeof(LzLutEntry)))) return rc;
                memcpy(lut_m4, M4, sizeof(M4)); lut_x = a->xdrop;
                LZ_HIP(hipMemcpyAsync(c.lut.p, lut_host.data(), lut_host.size() * sizeof(LzLutEntry), hipMemcpyHostToDevice, c.stream));
                LZ_HIP(hipStreamSynchronize(c.stream));
            }
        }
    }
    c.last_scan_mode = mode;

    std::vector<lz_hsp> plain;
    // ---- 3. per chunk, a three-stage pipeline over three streams and two sets of every per-chunk buffer:
    //   stream  (F): k_fill_hits -> k_hist + scans            keys, partition offsets           memory-bound
    //   stream3 (S): k_scan_hits -> k_scan_tasks              phase A, the 4-byte summaries     VALU-bound
    //   stream2 (B): k_partition -> k_settle                  records in partitions, phase B    latency-bound
    // so that chunk c's phase B, chunk c+1's scans and chunk c+2's fill share the CUs (the scan kernel leaves
    // LDS, registers and wave slots for the others' workgroups).  Phase B launches are ordered among themselves
    // on stream2 (diagEnd carries from chunk to chunk); a buffer set is rewritten only after its last reader.
    if (a->extend && max_chunk) for (int k = 0; k < nsets; k++) if ((rc = lzk_scan_reserve(c, k, mode, max_chunk))) return rc;
    LZ_HIP(hipEventRecord(c.ev_init, c.stream));              // state resets above are on stream 1
    LZ_HIP(hipStreamWaitEvent(c.stream2, c.ev_init, 0));
    LZ_HIP(hipStreamWaitEvent(c.stream3, c.ev_init, 0));
    size_t ci = 0;
    // Measured on the bench pair: 219-223 ms per step with the pipeline, 226 ms with everything on one stream --
    // the scan kernel runs at 75-85 % of the VALU issue rate, ~75 % LDS-pipe occupancy and 3 TB/s of 64-byte
    // sector fetches at once, so co-resident kernels mostly take turns with it.  The pipeline therefore is opt-in
    // (LZGPU_OVERLAP=1): by default the stage runs on one stream with one buffer set (5 GiB less to allocate,
    // per-kernel event times that mean what they say).
    const bool serial = !overlap;
    hipStream_t sF = c.stream, sS = serial ? c.stream : c.stream3, sB = serial ? c.stream : c.stream2;
    // phase B needs whole CUs (128 VGPRs x 1024 lanes) and cannot share one with the scan kernel: it runs on the
    // scans' stream, behind the NEXT chunk's scans, by which time its partition (which does overlap them) is done
    auto settle = [&](int set) -> int {
        LZ_HIP(hipStreamWaitEvent(sS, c.ev_part[set], 0));
        int r = lzk_settle(c, P, c.recs[set].as<u64>(), c.bin_base[set].as<u32>(), c.diag_end.as<u32>(), c.score_tab.as<s32>(),
                           c.hsp_out.as<LzHspRec>(), c.hsp_count.as<u32>(), out_cap, d_counters, sS);
        if (r) return r;
        LZ_HIP(hipEventRecord(c.ev_extended[set], sS));
        return 0;
    };
    int pending = -1;                                           // set whose phase B is still to be launched
    for (auto& ch : chunks) {
        const int set = (int)(ci % (size_t)nsets);
        if (!a->extend) {                                       // process_for_plain_hit: report every hit
            if ((rc = lzk_fill_hits(c, lo, ch.i0, ch.i1, c.wsk.as<u32>(), c.wsv.as<u32>(), n, c.off.as<u64>(), ch.base, c.keys[0].as<u64>(), sF))) return rc;
            std::vector<u64> hk(ch.nh);
            LZ_HIP(hipMemcpyAsync(hk.data(), c.keys[0].p, (size_t)ch.nh * 8, hipMemcpyDeviceToHost, c.stream));
            LZ_HIP(hipStreamSynchronize(c.stream));
            for (u64 k : hk) { u32 p2 = (u32)k; plain.push_back({ p2 + (u32)(k >> 32), p2, L, 0 }); }
            continue;
        }
        const bool reuse = ci >= (size_t)nsets;                 // the set has been through the pipeline before
        if (reuse && pending == set) { if ((rc = settle(pending))) return rc; pending = -1; }   // (fewer than three sets)
        // F: keys + histogram (every buffer of the set is free once its phase B is done)
        if (reuse) LZ_HIP(hipStreamWaitEvent(sF, c.ev_extended[set], 0));
        if ((rc = lzk_fill_hits(c, lo, ch.i0, ch.i1, c.wsk.as<u32>(), c.wsv.as<u32>(), n, c.off.as<u64>(), ch.base, c.keys[set].as<u64>(), sF))) return rc;
        LZ_HIP(hipEventRecord(c.ev_keys[set], sF));
        // S: the scans
        LZ_HIP(hipStreamWaitEvent(sS, c.ev_keys[set], 0));
        if ((rc = lzk_scan_hits(c, set, mode, P, Q, c.keys[set].as<u64>(), ch.nh, c.score_tab.as<s32>(), c.lut.as<LzLutEntry>(), c.bins[set].as<u8>(), sS))) return rc;
        LZ_HIP(hipEventRecord(c.ev_summ[set], sS));
        // B: the partition
        LZ_HIP(hipStreamWaitEvent(sB, c.ev_summ[set], 0));
        // (the tile histograms from the partition bytes the scan kernel left)
        if ((rc = lzk_hist(c, c.bins[set].as<u8>(), ch.nh, c.hist[set].as<u32>(), c.hist_part[set].as<u32>(), c.bin_base[set].as<u32>(), sB))) return rc;
        if ((rc = lzk_partition(c, set, c.keys[set].as<u64>(), ch.nh, c.hist[set].as<u32>(), c.hist_part[set].as<u32>(), c.recs[set].as<u64>(), sB))) return rc;
        LZ_HIP(hipEventRecord(c.ev_part[set], sB));
        // S again: phase B of the previous chunk (diagEnd carries from chunk to chunk: chunk order)
        if (pending >= 0) { if ((rc = settle(pending))) return rc; }
        pending = set;
        ci++;
    }
    if (pending >= 0) { if ((rc = settle(pending))) return rc; }
    g_hp.lap(3, "chunk loop launches");
    LZ_HIP(hipStreamSynchronize(c.stream3));
    LZ_HIP(hipStreamSynchronize(c.stream2));
    g_hp.lap(4, "wait for GPU");

    u64 hc[3] = { 0, 0, 0 }; u32 n_rec = 0;
    LZ_HIP(hipMemcpyAsync(hc, d_counters, 24, hipMemcpyDeviceToHost, c.stream));
    LZ_HIP(hipMemcpyAsync(&n_rec, c.hsp_count.p, 4, hipMemcpyDeviceToHost, c.stream));
    LZ_HIP(hipStreamSynchronize(c.stream));
    const bool gpu_counts = a->extend && n_rec > 0 && n_rec <= (u32)std::min<u64>(c.hsp_capacity, 0xFFFFFFF0ull);
    if (gpu_counts) {
        if ((rc = c.hsp_mc.ensure((size_t)n_rec * 20))) return rc;
        if ((rc = lzk_hsp_match_counts(c, c.hsp_out.as<LzHspRec>(), c.hsp_count.as<u32>(), n_rec, n_rec,
                                       c.target.raw_base(), qs->raw_base(), c.target.code_base(), qs->code_base(),
                                       c.hsp_mc.as<u32>(), c.stream))) return rc;
    }
    c.timer.resolve();
    { std::lock_guard<std::mutex> lk(c.counters_m);
      c.counters.words += hc[2]; c.counters.raw_hits += total_hits;
      c.counters.extensions += hc[0]; c.counters.bp_extended += hc[1]; }

    if (!a->extend) {
        lz_hsp* res = (lz_hsp*)malloc((plain.size() ? plain.size() : 1) * sizeof(lz_hsp));
        if (!res) return lz_fail(LZGPU_ERR_OOM, "host malloc failed");
        if (!plain.empty()) memcpy(res, plain.data(), plain.size() * sizeof(lz_hsp));
        *out = res; *n_out = plain.size();
        return 0;
    }
    if (n_rec > out_cap) return LZGPU_NH_HSP_OVERFLOW;

    // ---- 4. host finish: discovery order, entropy, threshold
    std::vector<LzHspRec> recs(n_rec);
    std::vector<u32> mc;
    if (n_rec) LZ_HIP(hipMemcpyAsync(recs.data(), c.hsp_out.p, (size_t)n_rec * sizeof(LzHspRec), hipMemcpyDeviceToHost, c.stream));
    if (gpu_counts) { mc.resize((size_t)n_rec * 5); LZ_HIP(hipMemcpyAsync(mc.data(), c.hsp_mc.p, (size_t)n_rec * 20, hipMemcpyDeviceToHost, c.stream)); }
    LZ_HIP(hipStreamSynchronize(c.stream));
    c.timer.resolve();
    g_hp.lap(5, "copy candidates");
    std::vector<lz_hsp> fin;
    if ((rc = lzh_finish_hsps(recs.data(), n_rec, c.target.host.data(), qhost, c.seed, c.geom.char_to_bits,
                              a->hsp_threshold, a->entropic, fin, gpu_counts ? mc.data() : nullptr, &c.last_order)))
        return lz_fail(rc, "internal: candidate HSP is not on a seed hit");
    lz_hsp* res = (lz_hsp*)malloc((fin.size() ? fin.size() : 1) * sizeof(lz_hsp));
    if (!res) return lz_fail(LZGPU_ERR_OOM, "host malloc failed");
    if (!fin.empty()) memcpy(res, fin.data(), fin.size() * sizeof(lz_hsp));
    { std::lock_guard<std::mutex> lk(c.counters_m); c.counters.hsps += fin.size(); }
    g_hp.lap(6, "host finish (order+entropy)");
    *out = res; *n_out = fin.size();
    return 0;
}

// ------------------------------------------------------------------------------ B1 + B2 of many windows (N3)
int lzk_window_search(LzCtx& c, const LzExtendParams& P, const LzSeedDev& sd, const lz_window* wins, u32 n, const s32* score_tab_dev,
                      std::vector<lz_hsp>& out, std::vector<u32>& counts);     // window_kernels.hip

extern "C" int lzgpu_window_search(const lz_window_search_args* a, lz_hsp** out, uint64_t* n_out, uint32_t** counts)
{
    int rc = require_init(); if (rc) return rc;
    LzCtx& c = g_ctx;
    if (!a || !out || !n_out || !counts || !a->sub || !a->seed || !a->char_to_bits || (!a->windows && a->n_windows))
        return lz_fail(LZGPU_ERR_ARG, "null argument");
    *out = nullptr; *n_out = 0; *counts = nullptr;
    if (!c.target.have_raw || c.target.len != c.geom.tlen) return lz_fail(LZGPU_ERR_STATE, "no target on the device");
    LzSeedDev sd;
    if ((rc = lzh_seed_to_dev(a->seed, sd))) return rc;
    if (sd.nprobes != 1 || sd.weight > 14) return LZGPU_NH_UNSUPPORTED;
    SeqSlot* qs;
    if (a->query) {
        if (a->qlen >= 0x7FFFFFFFu) return LZGPU_NH_SIZE;
        qs = lz_query_slot(c, -1, true);
        if ((rc = slot_upload(c, *qs, a->query, a->qlen, false))) return rc;
    } else {
        qs = a->query_slot < 0 ? nullptr : lz_query_slot(c, a->query_slot, false);
        if (!qs) return lz_fail(LZGPU_ERR_ARG, "query slot %d is empty", a->query_slot);
    }
    for (u32 k = 0; k < a->n_windows; k++) {
        const lz_window& w = a->windows[k];
        if (w.t_len > 20480 || w.q_len > 20480) return LZGPU_NH_UNSUPPORTED;
        if ((u64)w.t_off + w.t_len > c.target.len || (u64)w.q_off + w.q_len > qs->len) return lz_fail(LZGPU_ERR_ARG, "window %u lies outside the sequences", k);
    }
    // score classes of the (masked) matrix and the code bytes of both sequences, as for the main search
    u8 rowc[256], colc[256], cls[256]; s32 tab[LZ_NCLASS * LZ_NCLASS];
    if ((rc = lzh_score_classes(a->sub, rowc, colc, tab))) return rc;
    DevBuf& wtab = c.win_tab;
    if ((rc = wtab.ensure(sizeof(tab)))) return rc;
    LZ_HIP(hipMemcpyAsync(wtab.p, tab, sizeof(tab), hipMemcpyHostToDevice, c.stream));
    LZ_HIP(hipStreamSynchronize(c.stream));
    lzh_make_cls(rowc, a->char_to_bits, cls);
    if ((rc = slot_encode(c, c.target, cls, g_cls_t))) return rc;
    lzh_make_cls(colc, a->char_to_bits, cls);
    if ((rc = slot_encode(c, *qs, cls, g_cls_q))) return rc;
    LzExtendParams P; memset(&P, 0, sizeof(P));
    P.tcode = c.target.code_base(); P.tlen = c.target.len; P.qcode = qs->code_base(); P.qlen = qs->len;
    P.xdrop = a->xdrop; P.min_score = a->hsp_threshold; P.seed_len = (u32)sd.length;
    std::vector<lz_hsp> hs; std::vector<u32> cn;
    if ((rc = lzk_window_search(c, P, sd, a->windows, a->n_windows, wtab.as<s32>(), hs, cn))) return rc;
    lz_hsp* res = (lz_hsp*)malloc((hs.size() ? hs.size() : 1) * sizeof(lz_hsp));
    u32* cnt = (u32*)malloc((cn.size() ? cn.size() : 1) * sizeof(u32));
    if (!res || !cnt) { free(res); free(cnt); return lz_fail(LZGPU_ERR_OOM, "host malloc failed"); }
    if (!hs.empty()) memcpy(res, hs.data(), hs.size() * sizeof(lz_hsp));
    if (!cn.empty()) memcpy(cnt, cn.data(), cn.size() * sizeof(u32));
    *out = res; *n_out = hs.size(); *counts = cnt;
    return 0;
}

// ------------------------------------------------------------------------------ instrumentation

extern "C" void lzgpu_counters_reset(void) { std::lock_guard<std::mutex> lk(g_ctx.counters_m); memset(&g_ctx.counters, 0, sizeof(g_ctx.counters)); }
extern "C" int  lzgpu_counters_get(lz_counters* out) { if (!out) return LZGPU_ERR_ARG; std::lock_guard<std::mutex> lk(g_ctx.counters_m); *out = g_ctx.counters; return 0; }
extern "C" void lzgpu_profile_enable(int enable) { g_ctx.timer.enabled = g_ctx.dp_timer.enabled = enable != 0; }
extern "C" void lzgpu_profile_reset(void) { g_ctx.timer.reset(); g_ctx.dp_timer.reset(); }
extern "C" int  lzgpu_profile_get(int n, const char** name, uint64_t* launches, double* total_ms)
{
    if (n >= (int)g_ctx.timer.names.size()) n -= (int)g_ctx.timer.names.size(); else if (n >= 0) n += 0x40000000;   // the seed stage's kernels first, then B3's
    KernelTimer& t = n >= 0x40000000 ? g_ctx.timer : g_ctx.dp_timer;
    n &= 0x3FFFFFFF;
    if (n < 0 || n >= (int)t.names.size()) return 1;
    if (name) *name = t.names[n].c_str();
    if (launches) *launches = t.launches[n];
    if (total_ms) *total_ms = t.ms[n];
    return 0;
}
extern "C" int lzgpu_set_hit_capacity(uint64_t n)
{
    if (n < 1024 || n > (1ull << 31)) return LZGPU_ERR_ARG;     // hit indices inside a chunk are 32-bit
    g_ctx.hit_capacity = n; return 0;
}
extern "C" int lzgpu_set_bucket_owner(uint32_t n_owners, uint32_t owner)
{
    if (n_owners < 1 || n_owners > LZ_DIAG_SIZE || owner >= n_owners) return LZGPU_ERR_ARG;
    g_ctx.n_owners = n_owners; g_ctx.owner = owner;
    return 0;
}
extern "C" int lzgpu_last_hsp_order(uint64_t* out, uint64_t n)
{
    if (!out || 2 * n != g_ctx.last_order.size()) return LZGPU_ERR_ARG;
    memcpy(out, g_ctx.last_order.data(), g_ctx.last_order.size() * 8);
    return 0;
}
extern "C" int lzgpu_last_scan_mode(void) { return g_ctx.last_scan_mode; }
extern "C" int lzgpu_set_scan_mode(int min_mode) { if (min_mode < 0 || min_mode > 2) return LZGPU_ERR_ARG; g_ctx.min_scan_mode = min_mode; return 0; }
extern "C" int lzgpu_set_hsp_capacity(uint64_t n) { if (n < 16) return LZGPU_ERR_ARG; g_ctx.hsp_capacity = n; return 0; }
