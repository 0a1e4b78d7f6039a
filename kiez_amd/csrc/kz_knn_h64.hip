// fp16 fused kernel with 64 queries per wave (kz_knn_h64.h), K' = 16: ordinary and dual-pass builds for 4 .. 13 slices.
#include "kz_common.h"
#include "kz_knn_device.h"
#include "kz_knn_h64.h"

template <int NSR, bool DUAL>
static int kz_h64_occ(int* blocks_per_cu, int lds_pad) {
    const void* kern = (const void*)kz_knn_cand_h64_kernel<NSR, DUAL>;
    const int lds = KzH64Cfg<NSR, DUAL>::LDS_BYTES + lds_pad;
    KZ_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int nb = 0;
    KZ_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, lds));
    *blocks_per_cu = nb < 1 ? 1 : nb;
    return KZ_OK;
}

template <int NSR, bool DUAL>
static int kz_h64_run(kz_ctx* ctx, const KnnCandParams& p, int n_blocks) {
    KnnCandParams pc = p;
    void* args[] = {&pc};
    KZ_HIP(hipLaunchKernel((const void*)kz_knn_cand_h64_kernel<NSR, DUAL>, dim3(n_blocks), dim3(256), args,
                           (size_t)(KzH64Cfg<NSR, DUAL>::LDS_BYTES + KZ_K_LDS_PAD), ctx->stream));
    return KZ_OK;
}

#define KZ_DISPATCH_H64(rc, fn, args, DUALV)        \
    do {                                            \
        switch (n_slices) {                         \
            case 4: rc = fn<4, DUALV> args; break;  \
            case 5: rc = fn<5, DUALV> args; break;  \
            case 6: rc = fn<6, DUALV> args; break;  \
            case 7: rc = fn<7, DUALV> args; break;  \
            case 8: rc = fn<8, DUALV> args; break;  \
            case 9: rc = fn<9, DUALV> args; break;  \
            case 10: rc = fn<10, DUALV> args; break; \
            case 11: rc = fn<11, DUALV> args; break; \
            case 12: rc = fn<12, DUALV> args; break; \
            case 13: rc = fn<13, DUALV> args; break; \
            default: rc = KZ_ERR_INVALID; break;    \
        }                                           \
    } while (0)

// slice counts this kernel is built for (kz_knn.hip asks before it plans a pass)
bool kz_h64_supports(int n_slices) { return n_slices >= 4 && n_slices <= 13; }

int kz_h64_occupancy(int n_slices, int dual, int* blocks_per_cu, int lds_pad) {
    int rc;
    if (dual)
        KZ_DISPATCH_H64(rc, kz_h64_occ, (blocks_per_cu, lds_pad), true);
    else
        KZ_DISPATCH_H64(rc, kz_h64_occ, (blocks_per_cu, lds_pad), false);
    return rc;
}

int kz_h64_launch(int n_slices, int dual, kz_ctx* ctx, const KnnCandParams& p, int n_blocks) {
    int rc;
    if (dual)
        KZ_DISPATCH_H64(rc, kz_h64_run, (ctx, p, n_blocks), true);
    else
        KZ_DISPATCH_H64(rc, kz_h64_run, (ctx, p, n_blocks), false);
    return rc;
}
