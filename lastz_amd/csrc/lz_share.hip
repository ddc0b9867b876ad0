// lz_share.hip -- lzgpu_table_share (include/lzgpu.h): the position table of rank 0 on every rank of a
// one-process-per-GPU run, for callers without a communicator of their own (the lastz binding).
// The data path is one ncclBroadcast per table buffer over RCCL / xGMI; librccl is loaded on first use so
// that single-GPU runs never touch it.  Rendezvous (geometry, the RCCL unique id) goes through small files
// in a directory all ranks see.  LZGPU_SHARE_TRANSPORT=file replaces the broadcast by host staging through
// that directory: for ranks that share ONE device (tests on a one-GPU box), where RCCL refuses to run.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <string>
#include <vector>
#include "lz_ctx.hpp"

namespace {
struct NcclUid { char internal[128]; };
typedef int (*fn_get_uid)(NcclUid*);
typedef int (*fn_init_rank)(void** comm, int nranks, NcclUid id, int rank);
typedef int (*fn_bcast)(const void* send, void* recv, size_t count, int dtype, int root, void* comm, hipStream_t s);
typedef int (*fn_destroy)(void* comm);
typedef const char* (*fn_errstr)(int);

// Rendezvous files start with the run's nonce (LZGPU_SHARE_NONCE, set by the launcher; 16 bytes, zero-padded): a
// file left behind by an earlier run in a reused directory is not adopted.  A rank that fails drops `abort` into
// the directory's parent (the launcher does the same when a rank dies): waiting ranks give up at once instead of
// sitting out the timeout (LZGPU_SHARE_TIMEOUT_S, default 600).
struct Nonce { char b[16]; };
Nonce run_nonce() { Nonce n; memset(&n, 0, sizeof(n)); if (const char* e = getenv("LZGPU_SHARE_NONCE")) strncpy(n.b, e, sizeof(n.b)); return n; }
int wait_seconds() { const char* e = getenv("LZGPU_SHARE_TIMEOUT_S"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 600; }
bool write_file_atomic(const std::string& path, const void* p, size_t n)
{
    const std::string tmp = path + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const Nonce nn = run_nonce();
    const bool ok = fwrite(&nn, 1, sizeof(nn), f) == sizeof(nn) && fwrite(p, 1, n, f) == n;
    fclose(f);
    return ok && rename(tmp.c_str(), path.c_str()) == 0;
}
bool read_file_wait(const std::string& path, void* p, size_t n, int timeout_s)
{
    const Nonce want = run_nonce();
    const size_t cut = path.find_last_of('/');
    const std::string dir = cut == std::string::npos ? std::string(".") : path.substr(0, cut);
    const std::string abort1 = dir + "/abort", abort2 = dir + "/../abort";
    for (int t = 0; t < timeout_s * 100; t++) {
        FILE* f = fopen(path.c_str(), "rb");
        if (f) {
            Nonce got;
            const bool ok = fread(&got, 1, sizeof(got), f) == sizeof(got) && memcmp(&got, &want, sizeof(got)) == 0 && fread(p, 1, n, f) == n;
            fclose(f);
            if (ok) return true;
        }
        if ((t & 15) == 15 && (access(abort1.c_str(), F_OK) == 0 || access(abort2.c_str(), F_OK) == 0)) return false;
        usleep(10000);
    }
    return false;
}
void drop_abort_marker(const std::string& dir) { FILE* f = fopen((dir + "/../abort").c_str(), "wb"); if (f) fclose(f); }
}

static int table_share_impl(int rank, int world, const char* dir)
{
    LzCtx& c = lz_ctx();
    if (world <= 1 && !getenv("LZGPU_SHARE_FORCE")) return 0;   // (LZGPU_SHARE_FORCE: a one-rank run still goes through the transport -- tests)
    if (world < 1) world = 1;
    if (rank < 0 || rank >= world || !dir) return lz_fail(LZGPU_ERR_ARG, "lzgpu_table_share: bad rank / world / directory");
    { int rc = lz_bind_thread(); if (rc) return rc; }
    const std::string d(dir);
    const char* tr = getenv("LZGPU_SHARE_TRANSPORT");
    const bool by_file = tr && strcmp(tr, "file") == 0;
    const int wait_s = wait_seconds();
    int rc;

    // ---- geometry
    lz_table_geom g;
    if (rank == 0) {
        if ((rc = lzgpu_table_geom(&g))) return rc;
        if (!write_file_atomic(d + "/geom", &g, sizeof(g))) return lz_fail(LZGPU_ERR_STATE, "lzgpu_table_share: cannot write %s/geom", dir);
    } else {
        if (!read_file_wait(d + "/geom", &g, sizeof(g), wait_s)) return lz_fail(LZGPU_ERR_STATE, "lzgpu_table_share: no geometry from rank 0 in %s", dir);
        if ((rc = lzgpu_table_adopt(&g))) return rc;
    }
    void* ptr[3]; uint64_t bytes[3];
    if ((rc = lzgpu_table_buffers(ptr, bytes))) return rc;

    if (by_file) {
        for (int k = 0; k < 3; k++) {
            const std::string path = d + "/buf" + std::to_string(k);
            std::vector<char> host(bytes[k] ? bytes[k] : 1);
            if (rank == 0) {
                if (bytes[k]) LZ_HIP(hipMemcpy(host.data(), ptr[k], bytes[k], hipMemcpyDeviceToHost));
                if (!write_file_atomic(path, host.data(), bytes[k])) return lz_fail(LZGPU_ERR_STATE, "lzgpu_table_share: cannot write %s", path.c_str());
            } else {
                if (!read_file_wait(path, host.data(), bytes[k], wait_s)) return lz_fail(LZGPU_ERR_STATE, "lzgpu_table_share: %s did not arrive", path.c_str());
                if (bytes[k]) LZ_HIP(hipMemcpy(ptr[k], host.data(), bytes[k], hipMemcpyHostToDevice));
            }
        }
    } else {
        static void* lib = nullptr;
        static fn_get_uid get_uid; static fn_init_rank init_rank; static fn_bcast bcast; static fn_destroy destroy; static fn_errstr errstr;
        if (!lib) {
            lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!lib) return lz_fail(LZGPU_ERR_STATE, "lzgpu_table_share: librccl not found (%s)", dlerror());
            get_uid = (fn_get_uid)dlsym(lib, "ncclGetUniqueId"); init_rank = (fn_init_rank)dlsym(lib, "ncclCommInitRank");
            bcast = (fn_bcast)dlsym(lib, "ncclBroadcast"); destroy = (fn_destroy)dlsym(lib, "ncclCommDestroy"); errstr = (fn_errstr)dlsym(lib, "ncclGetErrorString");
            if (!get_uid || !init_rank || !bcast || !destroy) return lz_fail(LZGPU_ERR_STATE, "lzgpu_table_share: librccl lacks a symbol");
        }
        // RCCL announces itself on stdout: the caller's stdout is its output file (lastz writes alignments there)
        fflush(stdout);
        const int saved_out = dup(1);
        if (saved_out >= 0) (void)dup2(2, 1);
        struct Restore { int fd; ~Restore() { if (fd >= 0) { fflush(stdout); (void)dup2(fd, 1); close(fd); } } } restore{ saved_out };
        NcclUid id; int e;
        if (rank == 0) {
            if ((e = get_uid(&id))) return lz_fail(LZGPU_ERR_HIP, "ncclGetUniqueId: %s", errstr ? errstr(e) : "?");
            if (!write_file_atomic(d + "/nccl_id", &id, sizeof(id))) return lz_fail(LZGPU_ERR_STATE, "lzgpu_table_share: cannot write %s/nccl_id", dir);
        } else if (!read_file_wait(d + "/nccl_id", &id, sizeof(id), wait_s)) return lz_fail(LZGPU_ERR_STATE, "lzgpu_table_share: no RCCL id from rank 0");
        void* comm = nullptr;
        if ((e = init_rank(&comm, world, id, rank))) return lz_fail(LZGPU_ERR_HIP, "ncclCommInitRank: %s", errstr ? errstr(e) : "?");
        for (int k = 0; k < 3; k++)
            if (bytes[k] && (e = bcast(ptr[k], ptr[k], (size_t)bytes[k], /*ncclUint8*/ 1, 0, comm, c.stream))) { destroy(comm); return lz_fail(LZGPU_ERR_HIP, "ncclBroadcast: %s", errstr ? errstr(e) : "?"); }
        LZ_HIP(hipStreamSynchronize(c.stream));
        destroy(comm);
    }
    if (rank != 0 && (rc = lzgpu_table_commit())) return rc;
    return 0;
}

extern "C" int lzgpu_table_share(int rank, int world, const char* dir)
{
    const int rc = table_share_impl(rank, world, dir);
    if (rc != 0 && dir) drop_abort_marker(dir);                 // the other ranks stop waiting for this one
    return rc;
}

// ------------------------------------------------------------------------------------------------
// lzgpu_table_save / lzgpu_table_load: the same payload as a versioned file (the role of the reference's capsule
// files, src/capsule.c:44-..., in this library's own layout)
namespace {
struct TabFileHead { char magic[8]; uint32_t version, endian, geom_bytes, pad; uint64_t bytes[3]; };
const char kMagic[8] = { 'L', 'Z', 'G', 'P', 'U', 'T', 'A', 'B' };
uint64_t fnv_more(const void* p, size_t n, uint64_t h) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } return h; }
}

extern "C" int lzgpu_table_save(const char* path)
{
    LzCtx& c = lz_ctx();
    if (!path) return lz_fail(LZGPU_ERR_ARG, "lzgpu_table_save: null path");
    lz_table_geom g; int rc;
    if ((rc = lzgpu_table_geom(&g))) return rc;
    void* ptr[3]; uint64_t bytes[3];
    if ((rc = lzgpu_table_buffers(ptr, bytes))) return rc;
    if ((rc = lz_bind_thread())) return rc;
    TabFileHead h; memset(&h, 0, sizeof(h));
    memcpy(h.magic, kMagic, 8); h.version = 1; h.endian = 0x01020304u; h.geom_bytes = (uint32_t)sizeof(g);
    for (int k = 0; k < 3; k++) h.bytes[k] = bytes[k];
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return lz_fail(LZGPU_ERR_STATE, "lzgpu_table_save: cannot create %s", tmp.c_str());
    uint64_t sum = 1469598103934665603ull;
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(&g, sizeof(g), 1, f) == 1;
    sum = fnv_more(&h, sizeof(h), sum); sum = fnv_more(&g, sizeof(g), sum);
    std::vector<char> host;
    for (int k = 0; k < 3 && ok; k++) {
        host.resize(bytes[k] ? bytes[k] : 1);
        if (bytes[k] && hipMemcpy(host.data(), ptr[k], bytes[k], hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
        ok = bytes[k] == 0 || fwrite(host.data(), 1, bytes[k], f) == bytes[k];
        sum = fnv_more(host.data(), bytes[k], sum);
    }
    ok = ok && fwrite(&sum, 8, 1, f) == 1;
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), path) != 0) { unlink(tmp.c_str()); return lz_fail(LZGPU_ERR_STATE, "lzgpu_table_save: writing %s failed", path); }
    (void)c;
    return 0;
}

extern "C" int lzgpu_table_load(const char* path)
{
    LzCtx& c = lz_ctx();
    if (!path) return lz_fail(LZGPU_ERR_ARG, "lzgpu_table_load: null path");
    { int rc = lz_bind_thread(); if (rc) return rc; }
    FILE* f = fopen(path, "rb");
    if (!f) return lz_fail(LZGPU_ERR_ARG, "lzgpu_table_load: cannot open %s", path);
    TabFileHead h; lz_table_geom g;
    uint64_t sum = 1469598103934665603ull;
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, kMagic, 8) != 0 || h.version != 1 || h.endian != 0x01020304u
        || h.geom_bytes != sizeof(g) || fread(&g, sizeof(g), 1, f) != 1) { fclose(f); return lz_fail(LZGPU_ERR_ARG, "lzgpu_table_load: %s is not a version-1 table file", path); }
    sum = fnv_more(&h, sizeof(h), sum); sum = fnv_more(&g, sizeof(g), sum);
    int rc = lzgpu_table_adopt(&g);
    if (rc) { fclose(f); return rc; }
    void* ptr[3]; uint64_t bytes[3];
    if ((rc = lzgpu_table_buffers(ptr, bytes))) { fclose(f); return rc; }
    std::vector<char> host;
    for (int k = 0; k < 3; k++) {
        if (bytes[k] != h.bytes[k]) { fclose(f); return lz_fail(LZGPU_ERR_STATE, "lzgpu_table_load: %s: buffer %d has %llu bytes, its geometry says %llu", path, k, (unsigned long long)h.bytes[k], (unsigned long long)bytes[k]); }
        host.resize(bytes[k] ? bytes[k] : 1);
        if (bytes[k] && fread(host.data(), 1, bytes[k], f) != bytes[k]) { fclose(f); return lz_fail(LZGPU_ERR_STATE, "lzgpu_table_load: %s is truncated", path); }
        sum = fnv_more(host.data(), bytes[k], sum);
        if (bytes[k]) LZ_HIP(hipMemcpy(ptr[k], host.data(), bytes[k], hipMemcpyHostToDevice));
    }
    uint64_t want = 0;
    const bool ok = fread(&want, 8, 1, f) == 1 && want == sum;
    fclose(f);
    if (!ok) { c.have_table = false; return lz_fail(LZGPU_ERR_STATE, "lzgpu_table_load: %s fails its checksum", path); }
    return lzgpu_table_commit();
}
