// Context, memory and error plumbing of libkiez_amd.so.
#include "kz_common.h"
#include "kz_options.h"

#include <cstdlib>

static thread_local char g_err[512] = "";

void kz_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

int kz_abi_version(void) { return KZ_ABI_VERSION; }

const char* kz_last_error(void) { return g_err; }

int kz_device_count(int* n) {
    KZ_REQUIRE(n != nullptr, "kz_device_count: null output");
    KZ_HIP(hipGetDeviceCount(n));
    return KZ_OK;
}

int kz_ctx_create(int device, void* stream, kz_ctx** out) {
    KZ_REQUIRE(out != nullptr, "kz_ctx_create: null output");
    int n = 0;
    KZ_HIP(hipGetDeviceCount(&n));
    KZ_REQUIRE(device >= 0 && device < n, "kz_ctx_create: device %d out of range (%d visible)", device, n);
    KZ_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    KZ_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        kz_set_error("kz_ctx_create: device %d is %s; this library is built for gfx950 (MI355X) only", device,
                     prop.gcnArchName);
        return KZ_ERR_UNSUPPORTED;
    }
    kz_ctx* c = new kz_ctx();
    memset(c, 0, sizeof(*c));
    c->device = device;
    kz_options_defaults(c);   // (kz_options.h: the one table of options and their defaults)
    if (const char* pv = getenv("KZ_PRECISION"))  // A/B runs of the test-suite: fp32 | bf16 | fp16
        c->precision = (strcmp(pv, "fp32") == 0 || strcmp(pv, "1") == 0) ? 1 : ((strcmp(pv, "bf16") == 0 || strcmp(pv, "2") == 0) ? 2 : 0);
    c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (stream) {
        c->stream = (hipStream_t)stream;
        c->own_stream = false;
    } else {
        KZ_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    KZ_HIP(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    for (int i = 0; i < 12; ++i) KZ_HIP(hipEventCreate(&c->ev[i]));
    KZ_HIP(hipMalloc((void**)&c->d_counters, 64 * sizeof(int)));
    KZ_HIP(hipHostMalloc((void**)&c->h_counters, 64 * sizeof(int), hipHostMallocDefault));
    KZ_HIP(hipMemsetAsync(c->d_counters, 0, 64 * sizeof(int), c->stream));
    KZ_HIP(hipStreamSynchronize(c->stream));
    *out = c;
    return KZ_OK;
}

int kz_ctx_destroy(kz_ctx* c) {
    if (!c) return KZ_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->floor_buf) (void)hipFree(c->floor_buf);
    for (int i = 0; i < c->pool_n; ++i) (void)hipFree(c->pool[i].ptr);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->h_counters) (void)hipHostFree(c->h_counters);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    for (int i = 0; i < 12; ++i) (void)hipEventDestroy(c->ev[i]);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2), (void)hipStreamDestroy(c->stream2);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return KZ_OK;
}

int kz_ctx_trim(kz_ctx* c) {
    KZ_REQUIRE(c != nullptr, "kz_ctx_trim: null context");
    KZ_HIP(hipSetDevice(c->device));
    KZ_HIP(hipStreamSynchronize(c->stream));
    if (c->stream2) KZ_HIP(hipStreamSynchronize(c->stream2));   // (kz_knn_dual's reverse chain: nothing it uses may be freed under it)
    for (int i = 0; i < c->pool_n; ++i) (void)hipFree(c->pool[i].ptr);
    c->pool_n = 0;
    c->pool_bytes = 0;
    if (c->floor_buf) {
        (void)hipFree(c->floor_buf);
        c->floor_buf = nullptr;
        c->floor_bytes = 0;
    }
    if (c->scratch) {
        (void)hipFree(c->scratch);
        c->scratch = nullptr;
        c->scratch_bytes = 0;
    }
    return KZ_OK;
}

int kz_ctx_sync(kz_ctx* c) {
    KZ_REQUIRE(c != nullptr, "kz_ctx_sync: null context");
    KZ_HIP(hipStreamSynchronize(c->stream));
    return KZ_OK;
}

int kz_ctx_set_option(kz_ctx* c, const char* name, double value) {
    KZ_REQUIRE(c && name, "kz_ctx_set_option: null argument");
    for (int i = 0; i < KZ_N_OPTIONS; ++i) {
        const KzOption& o = KZ_OPTIONS[i];
        if (strcmp(name, o.name) != 0) continue;
        bool ok = o.kind == KZ_OPT_BOOL || (value >= o.lo && value <= o.hi && value == value);
        if (o.flags & KZ_OPT_SET) {
            bool listed = false;
            const int n = o.n_allowed < 0 ? -o.n_allowed : o.n_allowed;
            for (int j = 0; j < n; ++j) listed = listed || value == o.allowed[j];
            ok = o.n_allowed < 0 ? (ok || listed) : listed;    // (< 0: the listed values in addition to the range)
        }
        if (ok && o.kind == KZ_OPT_INT && value != (double)(long long)value) ok = false;
        if (!ok) {
            kz_set_error("kz_ctx_set_option: %s = %g is not allowed (range [%g, %g]%s)", name, value, o.lo, o.hi,
                         (o.flags & KZ_OPT_SET) ? ", listed values" : "");
            return KZ_ERR_INVALID;
        }
        kz_option_store(c, o, value);
        return KZ_OK;
    }
    kz_set_error("kz_ctx_set_option: unknown option '%s'", name);
    return KZ_ERR_INVALID;
}

int kz_malloc(kz_ctx* c, size_t bytes, void** d_ptr) {
    KZ_REQUIRE(c && d_ptr, "kz_malloc: null argument");
    KZ_HIP(hipSetDevice(c->device));
    return kz_pool_alloc(c, bytes, d_ptr);
}

int kz_free(kz_ctx* c, void* d_ptr) {
    KZ_REQUIRE(c != nullptr, "kz_free: null context");
    if (!d_ptr) return KZ_OK;
    KZ_HIP(hipSetDevice(c->device));
    kz_pool_free(c, d_ptr, 0);
    return KZ_OK;
}

int kz_memcpy_h2d(kz_ctx* c, void* d_dst, const void* h_src, size_t bytes) {
    KZ_REQUIRE(c && (bytes == 0 || (d_dst && h_src)), "kz_memcpy_h2d: null argument");
    if (bytes == 0) return KZ_OK;
    KZ_HIP(hipSetDevice(c->device));
    KZ_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
    KZ_HIP(hipStreamSynchronize(c->stream));
    return KZ_OK;
}

int kz_memcpy_d2h(kz_ctx* c, void* h_dst, const void* d_src, size_t bytes) {
    KZ_REQUIRE(c && (bytes == 0 || (h_dst && d_src)), "kz_memcpy_d2h: null argument");
    if (bytes == 0) return KZ_OK;
    KZ_HIP(hipSetDevice(c->device));
    KZ_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    KZ_HIP(hipStreamSynchronize(c->stream));
    return KZ_OK;
}

int kz_memcpy_d2d(kz_ctx* c, void* d_dst, const void* d_src, size_t bytes) {
    KZ_REQUIRE(c && (bytes == 0 || (d_dst && d_src)), "kz_memcpy_d2d: null argument");
    if (bytes == 0) return KZ_OK;
    KZ_HIP(hipSetDevice(c->device));
    KZ_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return KZ_OK;
}

}  // extern "C"

// ---- stream-ordered buffer pool ---------------------------------------------------------------------
// The work of a context runs on ONE stream, so handing a released buffer to the next allocation of a similar size is
// ordered behind its last use; this avoids hipFree (device-wide sync) + hipMalloc on every fit().  The one exception is
// kz_knn_dual's reverse chain on the second stream: kz_knn_dual holds every buffer that chain uses until it has synchronised
// stream2 itself; everything here that FREES device memory (eviction, the retry after a failed hipMalloc, a scratch block that
// grows, kz_ctx_trim) waits for BOTH streams first, so that no helper called while the chain is in flight can pull memory from
// under it.
static void kz_sync_streams(kz_ctx* c) {
    (void)hipStreamSynchronize(c->stream);
    if (c->stream2 && c->stream2 != c->stream) (void)hipStreamSynchronize(c->stream2);
}
static const size_t KZ_POOL_MAX_BYTES = (size_t)48 << 30;   // of 288 GB

static void kz_live_add(kz_ctx* c, void* ptr, size_t bytes) {
    for (int i = 0; i < KZ_LIVE_MAX; ++i)
        if (!c->live_ptr[i]) {
            c->live_ptr[i] = ptr;
            c->live_bytes[i] = bytes;
            return;
        }
}

static size_t kz_live_take(kz_ctx* c, void* ptr) {
    for (int i = 0; i < KZ_LIVE_MAX; ++i)
        if (c->live_ptr[i] == ptr) {
            c->live_ptr[i] = nullptr;
            return c->live_bytes[i];
        }
    return 0;  // untracked (table was full): plain hipFree
}

int kz_pool_alloc(kz_ctx* c, size_t bytes, void** out) {
    *out = nullptr;
    if (bytes == 0) bytes = 16;
    const size_t need = (bytes + 255) & ~(size_t)255;
    int best = -1;
    for (int i = 0; i < c->pool_n; ++i) {
        if (c->pool[i].bytes >= need && c->pool[i].bytes <= need + (need >> 3) &&
            (best < 0 || c->pool[i].bytes < c->pool[best].bytes))
            best = i;
    }
    if (best >= 0) {
        *out = c->pool[best].ptr;
        kz_live_add(c, c->pool[best].ptr, c->pool[best].bytes);
        c->pool_bytes -= c->pool[best].bytes;
        for (int i = best + 1; i < c->pool_n; ++i) c->pool[i - 1] = c->pool[i];   // (keeps the entries in release order)
        --c->pool_n;
        return KZ_OK;
    }
    void* base = nullptr;
    hipError_t e = hipMalloc(&base, need);
    if (e != hipSuccess) {  // release the cache and retry once
        kz_sync_streams(c);
        for (int i = 0; i < c->pool_n; ++i) (void)hipFree(c->pool[i].ptr);
        c->pool_n = 0;
        c->pool_bytes = 0;
        e = hipMalloc(&base, need);
    }
    if (e != hipSuccess) {
        kz_set_error("device allocation of %zu bytes failed: %s", need, hipGetErrorString(e));
        return KZ_ERR_NOMEM;
    }
    kz_live_add(c, base, need);
    *out = base;
    return KZ_OK;
}

void kz_pool_free(kz_ctx* c, void* ptr, size_t /*bytes*/) {
    if (!ptr) return;
    const size_t cap = kz_live_take(c, ptr);
    if (cap > 0 && cap <= KZ_POOL_MAX_BYTES) {
        // keep the buffer just released (the next call of the same shape asks for it again); when the cache is full the OLDEST
        // entries go -- a cache that refuses new buffers once stale ones fill it turns every call into hipMalloc + hipFree of
        // gigabytes (seen: 260 ms per fit after other workloads had run in the same process)
        bool synced = false;
        while (c->pool_n > 0 && (c->pool_n >= KZ_POOL_SLOTS || c->pool_bytes + cap > KZ_POOL_MAX_BYTES)) {
            if (!synced) kz_sync_streams(c);
            synced = true;
            (void)hipFree(c->pool[0].ptr);
            c->pool_bytes -= c->pool[0].bytes;
            for (int i = 1; i < c->pool_n; ++i) c->pool[i - 1] = c->pool[i];
            --c->pool_n;
        }
        c->pool[c->pool_n].ptr = ptr;
        c->pool[c->pool_n].bytes = cap;
        ++c->pool_n;
        c->pool_bytes += cap;
        return;
    }
    kz_sync_streams(c);
    (void)hipFree(ptr);
}

// the context's floor buffer (grown on demand, released with the context / kz_ctx_trim)
int kz_floor_buf(kz_ctx* c, size_t bytes, float** out) {
    if (bytes > c->floor_bytes) {
        kz_sync_streams(c);
        if (c->floor_buf) (void)hipFree(c->floor_buf);
        c->floor_buf = nullptr;
        c->floor_bytes = 0;
        const size_t want = bytes + bytes / 4;
        const hipError_t e = hipMalloc((void**)&c->floor_buf, want);
        if (e != hipSuccess) {
            kz_set_error("floor buffer allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
            return KZ_ERR_NOMEM;
        }
        c->floor_bytes = want;
    }
    *out = c->floor_buf;
    return KZ_OK;
}

int kz_scratch(kz_ctx* c, size_t bytes, void** out) {
    if (bytes > c->scratch_bytes) {
        kz_sync_streams(c);
        if (c->scratch) KZ_HIP(hipFree(c->scratch));
        c->scratch = nullptr;
        c->scratch_bytes = 0;
        size_t want = bytes + (bytes >> 3) + (1 << 20);
        hipError_t e = hipMalloc(&c->scratch, want);
        if (e != hipSuccess) {
            kz_set_error("scratch allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
            return KZ_ERR_NOMEM;
        }
        c->scratch_bytes = want;
    }
    *out = c->scratch;
    return KZ_OK;
}
