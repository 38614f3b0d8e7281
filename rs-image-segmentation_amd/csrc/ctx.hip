// Context, workspace, communication hook and per-kernel timers of librsseg_hip.so.
#include <dlfcn.h>
#include <stdarg.h>

#include <chrono>
#include <mutex>

#include "common.h"

int rs_fail(rsseg_ctx *ctx, int code, const char *fmt, ...)
{
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

static thread_local char g_err[256] = "no context";

extern "C" const char *rsseg_version(void) { return "rsseg-hip 0.1 (gfx950)"; }

extern "C" const char *rsseg_last_error(const rsseg_ctx *ctx) { return ctx ? ctx->err : g_err; }

extern "C" int rsseg_ctx_create(int device, void *stream, rsseg_ctx **out)
{
    if (!out) return RSSEG_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        snprintf(g_err, sizeof(g_err), "no HIP device available (%s)", hipGetErrorString(e));
        return RSSEG_ERR_HIP;
    }
    if (device < 0 || device >= ndev) {
        snprintf(g_err, sizeof(g_err), "device %d out of range (have %d)", device, ndev);
        return RSSEG_ERR_INVALID;
    }
    rsseg_ctx *ctx = new rsseg_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess) {
        delete ctx;
        snprintf(g_err, sizeof(g_err), "hipSetDevice(%d) failed", device);
        return RSSEG_ERR_HIP;
    }
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            snprintf(g_err, sizeof(g_err), "hipStreamCreate failed");
            return RSSEG_ERR_HIP;
        }
        ctx->own_stream = true;
    }
    *out = ctx;
    return RSSEG_OK;
}

// ---- RCCL driven by the library itself -------------------------------------------------------------------------------
// librccl is not a link-time dependency: the host names the copy its process already uses (torch ships one in torch/lib)
// and the five entry points are bound with dlsym.  The declarations below restate rccl.h (ncclUniqueId: 128 opaque bytes
// passed BY VALUE; ncclInt64 = 4, ncclFloat32 = 7, ncclFloat64 = 8; ncclSum = 0, ncclMax = 2, ncclMin = 3).
struct rs_nccl_id {
    char internal[RSSEG_RCCL_ID_BYTES];
};
struct rccl_api {
    void *handle = nullptr;
    int (*GetUniqueId)(rs_nccl_id *) = nullptr;
    int (*CommInitRank)(void **comm, int nranks, rs_nccl_id id, int rank) = nullptr;
    int (*AllReduce)(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t stream) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    char path[512] = {0};
};
static rccl_api g_rccl;
static std::mutex g_rccl_mu;

static const char *rccl_load(const char *path)   // nullptr on success, else what failed
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.handle) return nullptr;
    static char why[640];
    const char *names[] = {path && *path ? path : nullptr, "librccl.so.1", "librccl.so"};
    void *h = nullptr;
    for (const char *nm : names) {
        if (!nm) continue;
        h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (h) { snprintf(g_rccl.path, sizeof(g_rccl.path), "%s", nm); break; }
    }
    if (!h) {
        snprintf(why, sizeof(why), "dlopen(librccl) failed: %s", dlerror());
        return why;
    }
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy || !g_rccl.GetErrorString) {
        snprintf(why, sizeof(why), "%s lacks an ncclGetUniqueId / CommInitRank / AllReduce / CommDestroy / GetErrorString symbol", g_rccl.path);
        return why;
    }
    g_rccl.handle = h;
    return nullptr;
}

static void rccl_release(rsseg_ctx *ctx)
{
    if (ctx->rccl_comm) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)g_rccl.CommDestroy(ctx->rccl_comm);
        ctx->rccl_comm = nullptr;
    }
    if (ctx->own_comm_buf && ctx->d_comm) (void)hipFree(ctx->d_comm);
    if (ctx->own_comm_buf) { ctx->d_comm = nullptr; ctx->comm_bytes = 0; ctx->own_comm_buf = false; }
}

// rsseg_allreduce_fn over the context's communicator: in place in the communication buffer, enqueued on the context's
// stream — ordered after the kernels that left their partials there and before the kernels that read the result, no host
// wait (the contract of include/rsseg.h).  torch's default stream reaches the library as hipStreamLegacy; RCCL is handed
// the null stream for it, the same stream under the name torch itself passes to RCCL.
static int rccl_allreduce(void *user, int64_t offset, int64_t count, int dtype, int op)
{
    rsseg_ctx *ctx = (rsseg_ctx *)user;
    static const int types[3] = {7 /* ncclFloat32 */, 8 /* ncclFloat64 */, 4 /* ncclInt64 */};
    static const int ops[3] = {0 /* ncclSum */, 3 /* ncclMin */, 2 /* ncclMax */};
    if (dtype < 0 || dtype > 2 || op < 0 || op > 2 || offset < 0 || count < 0) return 1;
    const size_t esz = dtype == RSSEG_F32 ? 4 : 8;
    if ((size_t)offset + esz * (size_t)count > ctx->comm_bytes) return 2;
    hipStream_t st = ctx->stream == hipStreamLegacy ? (hipStream_t) nullptr : ctx->stream;
    void *p = ctx->d_comm + offset;
    const int rc = g_rccl.AllReduce(p, p, (size_t)count, types[dtype], ops[op], ctx->rccl_comm, st);
    if (rc != 0) {
        snprintf(ctx->err, sizeof(ctx->err), "ncclAllReduce: %s", g_rccl.GetErrorString(rc));
        return 3;
    }
    return 0;
}

extern "C" int rsseg_rccl_unique_id(const char *librccl_path, void *id_out)
{
    if (!id_out) return RSSEG_ERR_INVALID;
    if (const char *why = rccl_load(librccl_path)) {
        snprintf(g_err, sizeof(g_err), "%s", why);
        return RSSEG_ERR_COMM;
    }
    rs_nccl_id id;
    const int rc = g_rccl.GetUniqueId(&id);
    if (rc != 0) {
        snprintf(g_err, sizeof(g_err), "ncclGetUniqueId: %s", g_rccl.GetErrorString(rc));
        return RSSEG_ERR_COMM;
    }
    memcpy(id_out, id.internal, RSSEG_RCCL_ID_BYTES);
    return RSSEG_OK;
}

extern "C" int rsseg_ctx_set_comm_rccl(rsseg_ctx *ctx, int rank, int world, const void *unique_id, const char *librccl_path, void *d_comm,
                                       size_t comm_bytes)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (world < 1 || rank < 0 || rank >= world || world > RSSEG_MAX_RANKS) return rs_fail(ctx, RSSEG_ERR_INVALID, "bad rank/world %d/%d", rank, world);
    if (!unique_id) return rs_fail(ctx, RSSEG_ERR_INVALID, "set_comm_rccl: no unique id (rank 0: rsseg_rccl_unique_id, then hand the 128 bytes to every rank)");
    if (d_comm && comm_bytes < (1u << 20)) return rs_fail(ctx, RSSEG_ERR_INVALID, "set_comm_rccl: the communication buffer must hold >= 1 MiB");
    if (const char *why = rccl_load(librccl_path)) return rs_fail(ctx, RSSEG_ERR_COMM, "%s", why);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    rccl_release(ctx);
    if (ctx->h_comm) (void)hipHostFree(ctx->h_comm);
    ctx->h_comm = nullptr;
    ctx->comm_on = false;
    ctx->allreduce = nullptr;
    if (!d_comm) {
        comm_bytes = (size_t)4 << 20;
        HIPCHK(ctx, hipMalloc(&d_comm, comm_bytes));
        HIPCHK(ctx, hipMemset(d_comm, 0, comm_bytes));
        ctx->own_comm_buf = true;
    }
    ctx->d_comm = (char *)d_comm;
    ctx->comm_bytes = comm_bytes;
    rs_nccl_id id;
    memcpy(id.internal, unique_id, RSSEG_RCCL_ID_BYTES);
    void *comm = nullptr;
    const int rc = g_rccl.CommInitRank(&comm, world, id, rank);     // collective: returns when all `world` ranks have called it
    if (rc != 0 || !comm) {
        rccl_release(ctx);
        return rs_fail(ctx, RSSEG_ERR_COMM, "ncclCommInitRank(rank %d of %d) on %s: %s", rank, world, g_rccl.path, g_rccl.GetErrorString(rc));
    }
    ctx->rccl_comm = comm;
    ctx->rank = rank;
    ctx->world = world;
    ctx->allreduce = rccl_allreduce;
    ctx->comm_user = ctx;
    ctx->comm_on = true;     // also with world == 1: identity reductions through RCCL (how a one-GPU box exercises this path)
    if (hipHostMalloc((void **)&ctx->h_comm, comm_bytes, hipHostMallocDefault) != hipSuccess) {
        rccl_release(ctx);
        ctx->comm_on = false;
        ctx->allreduce = nullptr;
        return rs_fail(ctx, RSSEG_ERR_NOMEM, "hipHostMalloc(%zu) for the communication staging buffer failed", comm_bytes);
    }
    return RSSEG_OK;
}

extern "C" void rsseg_ctx_destroy(rsseg_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->prof)
        for (auto &p : kv.second.pending) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
    for (auto ev : ctx->event_pool) (void)hipEventDestroy(ev);
    if (ctx->d_ws) (void)hipFree(ctx->d_ws);
    if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
    if (ctx->h_comm) (void)hipHostFree(ctx->h_comm);
    rccl_release(ctx);
    if (ctx->d_mm) (void)hipFree(ctx->d_mm);
    if (ctx->forest.d_nodes) (void)hipFree(ctx->forest.d_nodes);
    if (ctx->forest.d_leafval) (void)hipFree(ctx->forest.d_leafval);
    if (ctx->forest.d_treeoff) (void)hipFree(ctx->forest.d_treeoff);
    if (ctx->forest.d_groups) (void)hipFree(ctx->forest.d_groups);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int rsseg_ctx_set_comm(rsseg_ctx *ctx, int rank, int world, rsseg_allreduce_fn fn, void *user,
                                  void *d_comm, size_t comm_bytes)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    rccl_release(ctx);
    if (world < 1 || rank < 0 || rank >= world || world > RSSEG_MAX_RANKS)
        return rs_fail(ctx, RSSEG_ERR_INVALID, "bad rank/world %d/%d", rank, world);
    if (world > 1 && (!fn || !d_comm || comm_bytes < (1u << 20)))
        return rs_fail(ctx, RSSEG_ERR_INVALID, "world > 1 needs an all-reduce hook and a >= 1 MiB comm buffer");
    ctx->rank = rank;
    ctx->world = world;
    ctx->allreduce = fn;
    ctx->comm_user = user;
    ctx->d_comm = (char *)d_comm;
    ctx->comm_bytes = comm_bytes;
    ctx->comm_on = world > 1 || (fn && d_comm && comm_bytes >= (1u << 20));
    if (ctx->h_comm) (void)hipHostFree(ctx->h_comm);
    ctx->h_comm = nullptr;
    if (ctx->comm_on) {
        HIPCHK(ctx, hipSetDevice(ctx->device));
        if (hipHostMalloc((void **)&ctx->h_comm, comm_bytes, hipHostMallocDefault) != hipSuccess)
            return rs_fail(ctx, RSSEG_ERR_NOMEM, "hipHostMalloc(%zu) for the communication staging buffer failed", comm_bytes);
    }
    return RSSEG_OK;
}

extern "C" int rsseg_ctx_allreduce(rsseg_ctx *ctx, int64_t offset, int64_t count, int dtype, int op)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (!ctx->comm_on) return RSSEG_OK;     // one rank, nothing installed: the identity
    if (dtype < RSSEG_F32 || dtype > RSSEG_I64 || op < RSSEG_SUM || op > RSSEG_MAX || offset < 0 || count < 0 ||
        (size_t)offset + (dtype == RSSEG_F32 ? 4 : 8) * (size_t)count > ctx->comm_bytes)
        return rs_fail(ctx, RSSEG_ERR_INVALID, "allreduce: bad dtype / op / range");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int rc = ctx->allreduce(ctx->comm_user, offset, count, dtype, op);
    if (rc != 0) {
        char why[sizeof(ctx->err)];      // the native provider leaves RCCL's message in ctx->err: copy before formatting into it
        snprintf(why, sizeof(why), "%s", ctx->rccl_comm ? ctx->err : "");
        return rs_fail(ctx, RSSEG_ERR_COMM, "all-reduce returned %d%s%.400s", rc, why[0] ? ": " : "", why);
    }
    return RSSEG_OK;
}

int ws_reserve(rsseg_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->ws_bytes) return RSSEG_OK;
    HIPCHK(ctx, rs_sync(ctx));
    if (ctx->d_ws) HIPCHK(ctx, hipFree(ctx->d_ws));
    ctx->d_ws = nullptr;
    ctx->ws_bytes = 0;
    size_t want = (bytes + (1u << 20)) & ~((size_t)(1u << 20) - 1);
    hipError_t e = hipMalloc((void **)&ctx->d_ws, want);
    if (e != hipSuccess) return rs_fail(ctx, RSSEG_ERR_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    ctx->ws_bytes = want;
    return RSSEG_OK;
}

int pin_reserve(rsseg_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->pin_bytes) return RSSEG_OK;
    HIPCHK(ctx, rs_sync(ctx));
    if (ctx->h_pin) HIPCHK(ctx, hipHostFree(ctx->h_pin));
    ctx->h_pin = nullptr;
    ctx->pin_bytes = 0;
    size_t want = (bytes + 65536) & ~((size_t)65535);
    hipError_t e = hipHostMalloc((void **)&ctx->h_pin, want, hipHostMallocDefault);
    if (e != hipSuccess) return rs_fail(ctx, RSSEG_ERR_NOMEM, "hipHostMalloc(%zu) failed", want);
    ctx->pin_bytes = want;
    return RSSEG_OK;
}

int stream_sync(rsseg_ctx *ctx)
{
    if (ctx->async_mode) return RSSEG_OK;
    HIPCHK(ctx, rs_sync(ctx));
    return RSSEG_OK;
}

extern "C" int rsseg_ctx_set_async(rsseg_ctx *ctx, int on)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    ctx->async_mode = on != 0;
    return RSSEG_OK;
}

extern "C" int rsseg_ctx_sync(rsseg_ctx *ctx)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, rs_sync(ctx));
    return RSSEG_OK;
}

int comm_allreduce_host(rsseg_ctx *ctx, void *host, int64_t count, int dtype, int op)
{
    if (!ctx->comm_on) return RSSEG_OK;
    const size_t esz = dtype == RSSEG_F32 ? 4 : 8;
    const size_t bytes = esz * (size_t)count;
    if (bytes > ctx->comm_bytes) return rs_fail(ctx, RSSEG_ERR_COMM, "comm buffer too small (%zu > %zu)", bytes, ctx->comm_bytes);
    const auto t0 = std::chrono::steady_clock::now();
    // ONE host synchronisation per collective: pinned staging -> device buffer (async), the hook enqueues the
    // collective ordered after the context's stream and leaves its result visible to later work on that stream
    // (torch.distributed: the collective's stream waits for the current stream, the current stream for the collective),
    // device buffer -> pinned staging (async), then the single wait.
    memcpy(ctx->h_comm, host, bytes);
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_comm, ctx->h_comm, bytes, hipMemcpyHostToDevice, ctx->stream));
    int rc = ctx->allreduce(ctx->comm_user, 0, count, dtype, op);
    if (rc != 0) return rs_fail(ctx, RSSEG_ERR_COMM, "all-reduce hook returned %d", rc);
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_comm, ctx->d_comm, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, rs_sync(ctx));
    memcpy(host, ctx->h_comm, bytes);
    if (ctx->prof_on) {  // host wall time of the whole exchange (staging copies + collective), name "allreduce"
        prof_entry &e = ctx->prof["allreduce"];
        e.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        e.launches++;
    }
    return RSSEG_OK;
}

// ---- extrema of produced planes ------------------------------------------------------------
extern "C" int rsseg_ctx_collect_minmax(rsseg_ctx *ctx, int on)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (on && !ctx->d_mm) HIPCHK(ctx, hipMalloc((void **)&ctx->d_mm, sizeof(uint32_t) * 2 * RSSEG_MM_PLANES * RSSEG_MM_REPL));
    ctx->mm_collect = on != 0;
    ctx->mm_count = 0;
    return RSSEG_OK;
}

extern "C" int rsseg_ctx_last_minmax(rsseg_ctx *ctx, int plane, double *mn, double *mx)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (plane < 0 || plane >= ctx->mm_count) return rs_fail(ctx, RSSEG_ERR_INVALID, "last_minmax: plane %d of %d", plane, ctx->mm_count);
    if (mn) *mn = ctx->mm_min[plane];
    if (mx) *mx = ctx->mm_max[plane];
    return RSSEG_OK;
}

int mm_begin(rsseg_ctx *ctx, int nplanes)
{
    ctx->mm_count = 0;
    if (!ctx->mm_collect) return RSSEG_OK;
    // slot layout {min key, max key} per plane: 0xffffffff / 0 via two strided fills would need a kernel; one 2-D memset
    // does it without touching host memory (no synchronisation): bytes 0..3 of each 8-byte slot = 0xff, 4..7 = 0x00
    HIPCHK(ctx, hipMemsetAsync(ctx->d_mm, 0, sizeof(uint32_t) * 2 * RSSEG_MM_PLANES * RSSEG_MM_REPL, ctx->stream));
    HIPCHK(ctx, hipMemset2DAsync(ctx->d_mm, 8, 0xff, 4, RSSEG_MM_PLANES * RSSEG_MM_REPL, ctx->stream));
    (void)nplanes;
    return RSSEG_OK;
}

int mm_end(rsseg_ctx *ctx, int nplanes)
{
    if (!ctx->mm_collect) return RSSEG_OK;
    uint32_t kr[2 * RSSEG_MM_PLANES * RSSEG_MM_REPL], k[2 * RSSEG_MM_PLANES];
    HIPCHK(ctx, hipMemcpyAsync(kr, ctx->d_mm, sizeof(kr), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, rs_sync(ctx));
    for (int i = 0; i < RSSEG_MM_PLANES; i++) {   // the replicas of a slot: smallest min key, largest max key
        k[2 * i] = 0xffffffffu;
        k[2 * i + 1] = 0;
        for (int r = 0; r < RSSEG_MM_REPL; r++) {
            k[2 * i] = std::min(k[2 * i], kr[(r * RSSEG_MM_PLANES + i) * 2]);
            k[2 * i + 1] = std::max(k[2 * i + 1], kr[(r * RSSEG_MM_PLANES + i) * 2 + 1]);
        }
    }
    auto unkey = [](uint32_t key) {
        uint32_t u = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key;
        float f;
        memcpy(&f, &u, 4);
        return (double)f;
    };
    for (int i = 0; i < nplanes && i < RSSEG_MM_PLANES; i++) {
        ctx->mm_min[i] = unkey(k[2 * i]);      // an untouched slot decodes to NaN (keys 0xffffffff / 0): empty plane
        ctx->mm_max[i] = unkey(k[2 * i + 1]);
    }
    ctx->mm_count = nplanes < RSSEG_MM_PLANES ? nplanes : RSSEG_MM_PLANES;
    return RSSEG_OK;
}

// ---- profiling -----------------------------------------------------------------------------
static hipEvent_t get_event(rsseg_ctx *ctx)
{
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

prof_scope::prof_scope(rsseg_ctx *c, const char *name) : ctx(c)
{
    if (!c->prof_on) return;
    e = &c->prof[name];
    a = get_event(c);
    b = get_event(c);
    (void)hipEventRecord(a, c->stream);
}

prof_scope::~prof_scope()
{
    if (!e) return;
    (void)hipEventRecord(b, ctx->stream);
    e->pending.emplace_back(a, b);
    e->launches++;
}

void prof_retag(rsseg_ctx *ctx, const char *from, int count, const char *to)
{
    if (!ctx->prof_on || count <= 0) return;
    auto it = ctx->prof.find(from);
    if (it == ctx->prof.end()) return;
    prof_entry &src = it->second;
    prof_entry &dst = ctx->prof[to];
    while (count-- > 0 && !src.pending.empty()) {
        dst.pending.push_back(src.pending.back());
        src.pending.pop_back();
        src.launches--;
        dst.launches++;
    }
}

extern "C" int rsseg_ctx_host_syncs(rsseg_ctx *ctx, int reset, int64_t *count)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    if (count) *count = ctx->host_syncs;
    if (reset) ctx->host_syncs = 0;
    return RSSEG_OK;
}

static void prof_drain(rsseg_ctx *ctx)
{
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->prof) {
        for (auto &p : kv.second.pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) kv.second.ms += ms;
            ctx->event_pool.push_back(p.first);
            ctx->event_pool.push_back(p.second);
        }
        kv.second.pending.clear();
    }
}

extern "C" int rsseg_prof_enable(rsseg_ctx *ctx, int on)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    prof_drain(ctx);
    ctx->prof_on = on != 0;
    return RSSEG_OK;
}

extern "C" int rsseg_prof_reset(rsseg_ctx *ctx)
{
    if (!ctx) return RSSEG_ERR_INVALID;
    prof_drain(ctx);
    ctx->prof.clear();
    return RSSEG_OK;
}

extern "C" int rsseg_prof_get(rsseg_ctx *ctx, const char *name, double *total_ms, int64_t *launches)
{
    if (!ctx || !name) return RSSEG_ERR_INVALID;
    prof_drain(ctx);
    auto it = ctx->prof.find(name);
    if (total_ms) *total_ms = it == ctx->prof.end() ? 0.0 : it->second.ms;
    if (launches) *launches = it == ctx->prof.end() ? 0 : it->second.launches;
    return RSSEG_OK;
}

// ---- host-only helpers: TIFF LZW (TIFF 6.0 section 13; the variant GDAL / libtiff write and read) -----------------
// The reference asks rasterio for compress='lzw' (scripts/2_feature_extraction.py:239-258, scripts/3_classification.py:
// 509-538); rsseg/tiff.py calls these per strip / tile.  Codes are packed MSB first, 9..12 bits, ClearCode 256,
// EndOfInformation 257, the code width grows one code early ("early change"), the table is cleared at 4094 entries.
extern "C" int64_t rsseg_host_lzw_encode(const uint8_t *in, int64_t n, uint8_t *out, int64_t cap)
{
    if (!in || !out || n < 0 || cap < 16) return -1;
    // open-addressing hash of (prefix code, byte) -> code
    const int HSIZE = 9001;
    std::vector<int32_t> hkey(HSIZE, -1);
    std::vector<uint16_t> hval(HSIZE, 0);
    int64_t op = 0;
    uint64_t acc = 0;
    int nacc = 0, nbits = 9, free_ent = 258;
    auto put = [&](int code) -> bool {
        acc = (acc << nbits) | (uint64_t)code;
        nacc += nbits;
        while (nacc >= 8) {
            if (op >= cap) return false;
            out[op++] = (uint8_t)(acc >> (nacc - 8));
            nacc -= 8;
        }
        return true;
    };
    auto reset = [&]() {
        std::fill(hkey.begin(), hkey.end(), -1);
        free_ent = 258;
        nbits = 9;
    };
    if (!put(256)) return -2;
    if (n == 0) {
        if (!put(257)) return -2;
        if (nacc > 0) { if (op >= cap) return -2; out[op++] = (uint8_t)(acc << (8 - nacc)); }
        return op;
    }
    int ent = in[0];
    for (int64_t i = 1; i < n; i++) {
        const int c = in[i];
        const int32_t key = (ent << 8) | c;
        int h = (int)(((uint32_t)key * 2654435761u) % (uint32_t)HSIZE);
        bool found = false;
        while (hkey[h] != -1) {
            if (hkey[h] == key) { found = true; break; }
            if (++h == HSIZE) h = 0;
        }
        if (found) { ent = hval[h]; continue; }
        if (!put(ent)) return -2;
        ent = c;
        hkey[h] = key;
        hval[h] = (uint16_t)free_ent++;
        if (free_ent == 4094) {          // table full: ClearCode, start over
            if (!put(256)) return -2;
            reset();
        } else if (free_ent > (1 << nbits) - 1)
            nbits++;
    }
    if (!put(ent)) return -2;
    // libtiff's LZWPostEncode: the last code also counts toward the width of EndOfInformation
    free_ent++;
    if (free_ent == 4094) { if (!put(256)) return -2; nbits = 9; }
    else if (free_ent > (1 << nbits) - 1) nbits++;
    if (!put(257)) return -2;
    if (nacc > 0) { if (op >= cap) return -2; out[op++] = (uint8_t)(acc << (8 - nacc)); }
    return op;
}

extern "C" int64_t rsseg_host_lzw_decode(const uint8_t *in, int64_t n, uint8_t *out, int64_t cap)
{
    if (!in || !out || n < 0 || cap < 0) return -1;
    std::vector<uint16_t> prefix(4096, 0), length(4096, 0);
    std::vector<uint8_t> suffix(4096, 0), first(4096, 0);
    for (int i = 0; i < 256; i++) { suffix[i] = first[i] = (uint8_t)i; length[i] = 1; }
    int64_t ip = 0, op = 0;
    uint64_t acc = 0;
    int nacc = 0, nbits = 9, free_ent = 258, old = -1;
    auto get = [&]() -> int {
        while (nacc < nbits) {
            if (ip >= n) return 257;
            acc = (acc << 8) | in[ip++];
            nacc += 8;
        }
        const int code = (int)((acc >> (nacc - nbits)) & ((1u << nbits) - 1));
        nacc -= nbits;
        return code;
    };
    auto emit = [&](int code) -> bool {
        const int len = length[code];
        if (op + len > cap) return false;
        int c = code;
        for (int k = len - 1; k >= 0; k--) { out[op + k] = suffix[c]; c = prefix[c]; }
        op += len;
        return true;
    };
    for (;;) {
        int code = get();
        if (code == 257) break;
        if (code == 256) {
            free_ent = 258;
            nbits = 9;
            code = get();
            if (code == 257) break;
            if (code > 255) return -3;
            if (!emit(code)) return -2;
            old = code;
            continue;
        }
        if (old < 0) return -3;
        if (code < free_ent) {
            if (!emit(code)) return -2;
            if (free_ent < 4096) { prefix[free_ent] = (uint16_t)old; suffix[free_ent] = first[code]; first[free_ent] = first[old]; length[free_ent] = length[old] + 1; free_ent++; }
        } else if (code == free_ent && free_ent < 4096) {
            prefix[free_ent] = (uint16_t)old; suffix[free_ent] = first[old]; first[free_ent] = first[old]; length[free_ent] = length[old] + 1;
            free_ent++;
            if (!emit(code)) return -2;
        } else
            return -3;
        old = code;
        if (free_ent > (1 << nbits) - 2 && nbits < 12) nbits++;
    }
    return op;
}
