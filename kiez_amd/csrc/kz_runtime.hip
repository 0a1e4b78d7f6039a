// Context, memory and error plumbing of libkiez_amd.so.
#include "kz_common.h"

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
    c->eps_scale = 1.0;
    c->force_splits = 0;
    c->h_wps = 0;
    c->h_wide = 0;
    c->long_k = 1;
    c->min_splits = 1;
    c->dual_stride = 1;
    c->dual_deal = 1;
    c->dual_overlap = 1;
    c->dual_sample_short = 1;
    c->dual_short_main = 1;
    c->esc_short = 1;
    c->short_ord = 1;
    c->esc_bf = 1;
    c->dual_short_kp = 16;
    c->dual_short_extra = 48;   // (400k x 400k, k = 50, 40 clusters: rows searched again 27.9k at 16, 10.7k from 32 on; uniform data: no difference)
    c->dual_rev_long = 1;
    // (500k x nb, k = 50, uniform, tiles per range -> shared sweep without / with the route: 49: 33.7 / 39.9 ms, 65: 39.4 / 45.6,
    //  78: 45.3 / 52.5, 98: 53.0 / 56.9, 133: 66.7 / 66.5, 195: 92.5 / 88.7, 390 (C3): -6 %)
    c->dual_short_min_tiles = 128;
    c->short_ord_min_tiles = 48;
    c->dual_short_div = 5;   // (500k x 500k, k = 50, ms per step and rows searched again: 4: 171.5 / 14, 5: 165.7 / 206, 6: 166.9 / 905, 8: 169.5 / 8904)
    c->lds_pad = 0;
    c->h_q64 = 2;
    c->tier_probe = 1024;
    c->wide_lists = 32;
    c->wide_sel = 256;
    c->dual_rank = 0;
    c->probe_min_pairs = 5e10;
    c->list_floor = 1;
    c->fin_fast_div = 1;
    c->floor_probe = 1024;
    c->floor_margin = 1.3;
    c->precision = 0;
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
    if (strcmp(name, "eps_scale") == 0) {
        KZ_REQUIRE(value > 0, "eps_scale must be > 0");
        c->eps_scale = value;
    } else if (strcmp(name, "force_splits") == 0) {
        KZ_REQUIRE(value >= 0 && value <= 64, "force_splits must be in [0, 64]");
        c->force_splits = (int)value;
    } else if (strcmp(name, "dual_max_gb") == 0) {
        KZ_REQUIRE(value >= 0, "dual_max_gb must be >= 0");
        c->dual_max_gb = value;
    } else if (strcmp(name, "dual_short_main") == 0) {
        c->dual_short_main = value != 0 ? 1 : 0;
    } else if (strcmp(name, "qgroup") == 0) {
        KZ_REQUIRE(value >= 0 && value <= 4096, "kz_ctx_set_option: qgroup must be in [0, 4096]");
        c->qgroup = (int)value;
    } else if (strcmp(name, "short_ord") == 0) {
        c->short_ord = value != 0 ? 1 : 0;
    } else if (strcmp(name, "esc_bf") == 0) {
        c->esc_bf = value != 0 ? 1 : 0;
    } else if (strcmp(name, "esc_short") == 0) {
        c->esc_short = value != 0 ? 1 : 0;
    } else if (strcmp(name, "short_ord_min_tiles") == 0) {
        KZ_REQUIRE(value >= 1, "kz_ctx_set_option: short_ord_min_tiles must be >= 1");
        c->short_ord_min_tiles = (int)value;
    } else if (strcmp(name, "dual_short_min_tiles") == 0) {
        KZ_REQUIRE(value >= 1, "kz_ctx_set_option: dual_short_min_tiles must be >= 1");
        c->dual_short_min_tiles = (int)value;
    } else if (strcmp(name, "dual_rev_long") == 0) {
        c->dual_rev_long = value != 0 ? 1 : 0;
    } else if (strcmp(name, "dual_short_extra") == 0) {
        KZ_REQUIRE(value >= 1 && value <= 200, "kz_ctx_set_option: dual_short_extra must be in [1, 200]");
        c->dual_short_extra = (int)value;
    } else if (strcmp(name, "dual_short_kp") == 0) {
        KZ_REQUIRE(value == 16 || value == 32, "kz_ctx_set_option: dual_short_kp must be 16 or 32");
        c->dual_short_kp = (int)value;
    } else if (strcmp(name, "dual_short_div") == 0) {
        KZ_REQUIRE(value >= 1 && value <= 16, "kz_ctx_set_option: dual_short_div must be in [1, 16]");
        c->dual_short_div = (int)value;
    } else if (strcmp(name, "dual_sample_short") == 0) {
        c->dual_sample_short = value != 0 ? 1 : 0;
    } else if (strcmp(name, "dual_overlap") == 0) {
        c->dual_overlap = value != 0 ? 1 : 0;
    } else if (strcmp(name, "long_k") == 0) {
        c->long_k = value != 0 ? 1 : 0;
    } else if (strcmp(name, "h_wide") == 0) {
        KZ_REQUIRE(value == 0 || value == 1, "h_wide must be 0 or 1");
        c->h_wide = (int)value;
    } else if (strcmp(name, "h_wps") == 0) {
        KZ_REQUIRE(value == 0 || value == 2 || value == 3, "h_wps must be 0 (automatic), 2 or 3");
        c->h_wps = (int)value;
    } else if (strcmp(name, "chunk_rows") == 0) {
        KZ_REQUIRE(value >= 0 && value <= 1e9, "chunk_rows must be >= 0");
        c->chunk_rows = (int)value;
    } else if (strcmp(name, "precision") == 0) {
        KZ_REQUIRE(value == 0 || value == 1 || value == 2,
                   "precision must be 0 (fp16 first pass), 2 (split-bf16 first pass) or 1 (float32 operands only)");
        c->precision = (int)value;
    } else if (strcmp(name, "dual_stride") == 0) {
        KZ_REQUIRE(value >= 0 && value <= 64, "dual_stride must be 0 (no dual pass), 1 (automatic) or in [2, 64]");
        c->dual_stride = (int)value;
    } else if (strcmp(name, "dual_deal") == 0) {
        c->dual_deal = value != 0;
    } else if (strcmp(name, "dual_force") == 0) {
        c->dual_force = value != 0;
    } else if (strcmp(name, "lds_pad") == 0) {
        KZ_REQUIRE(value >= 0 && value <= 90000, "lds_pad must be in [0, 90000]");
        c->lds_pad = (int)value;
    } else if (strcmp(name, "fin_fast_div") == 0) {
        KZ_REQUIRE(value == 0 || value == 1, "fin_fast_div must be 0 or 1");
        c->fin_fast_div = (int)value;
    } else if (strcmp(name, "probe_min_pairs") == 0) {
        KZ_REQUIRE(value >= 0, "probe_min_pairs must be >= 0");
        c->probe_min_pairs = value;
    } else if (strcmp(name, "list_floor") == 0) {
        KZ_REQUIRE(value == 0 || value == 1, "list_floor must be 0 or 1");
        c->list_floor = (int)value;
    } else if (strcmp(name, "floor_probe") == 0) {
        KZ_REQUIRE(value >= 0 && value <= 65536, "floor_probe must be in [0, 65536]");
        c->floor_probe = (int)value;
    } else if (strcmp(name, "floor_margin") == 0) {
        KZ_REQUIRE(value >= 0 && value <= 1e6, "floor_margin must be in [0, 1e6]");
        c->floor_margin = value;
    } else if (strcmp(name, "dual_rank") == 0) {
        KZ_REQUIRE(value >= -1 && value <= 128, "dual_rank must be -1 (k + 1), 0 (automatic) or in [1, 128]");
        c->dual_rank = (int)value;
    } else if (strcmp(name, "wide_lists") == 0) {
        KZ_REQUIRE(value == 0 || (value >= 2 && value <= 32), "wide_lists must be 0 or in [2, 32]");
        c->wide_lists = (int)value;
    } else if (strcmp(name, "wide_sel") == 0) {
        KZ_REQUIRE(value >= 16 && value <= 512, "wide_sel must be in [16, 512]");
        c->wide_sel = (int)value;
    } else if (strcmp(name, "tier_probe") == 0) {
        KZ_REQUIRE(value >= 0 && value <= 65536, "tier_probe must be in [0, 65536]");
        c->tier_probe = (int)value;
    } else if (strcmp(name, "h_q64") == 0) {
        KZ_REQUIRE(value == 0 || value == 1 || value == 2, "h_q64 must be 0 (never), 1 (wherever built) or 2 (automatic)");
        c->h_q64 = (int)value;
    } else if (strcmp(name, "min_splits") == 0) {
        KZ_REQUIRE(value >= 1 && value <= 32, "min_splits must be in [1, 32]");
        c->min_splits = (int)value;
    } else {
        kz_set_error("kz_ctx_set_option: unknown option '%s'", name);
        return KZ_ERR_INVALID;
    }
    return KZ_OK;
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
        while (c->pool_n > 0 && (c->pool_n >= 64 || c->pool_bytes + cap > KZ_POOL_MAX_BYTES)) {
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
