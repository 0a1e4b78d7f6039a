// fp16 fused kernel on v_mfma_f32_16x16x32_f16 (kz_knn_hx.h), K' = 16: ordinary and dual-pass builds for 2 .. 24 slices; three
// workgroups per CU up to 13 slices (the stationary query operands fit 168 VGPRs), two beyond (or with wps = 2).
#include "kz_common.h"
#include "kz_knn_device.h"
#include "kz_knn_hx.h"

constexpr int KZ_HX_WPS3_MAX = 13;

template <int NSR, int WPS, bool DUAL>
static int kz_hx_occ(int* blocks_per_cu, int lds_pad) {
    const void* kern = (const void*)kz_knn_cand_hx_kernel<NSR, WPS, DUAL>;
    const int lds = KzHxCfg<NSR, WPS, DUAL>::LDS_BYTES + lds_pad;
    KZ_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int nb = 0;
    KZ_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, lds));
    *blocks_per_cu = nb < 1 ? 1 : nb;
    return KZ_OK;
}

template <int NSR, int WPS, bool DUAL>
static int kz_hx_run(kz_ctx* ctx, const KnnCandParams& p, int n_blocks) {
    KnnCandParams pc = p;
    void* args[] = {&pc};
    KZ_HIP(hipLaunchKernel((const void*)kz_knn_cand_hx_kernel<NSR, WPS, DUAL>, dim3(n_blocks), dim3(256), args,
                           (size_t)(KzHxCfg<NSR, WPS, DUAL>::LDS_BYTES + ctx->lds_pad), ctx->stream));
    return KZ_OK;
}

template <int NSR, bool DUAL>
static int kz_hx_occ_w(int wps, int* blocks_per_cu, int lds_pad) {
    if constexpr (NSR <= KZ_HX_WPS3_MAX) {
        if (wps != 2) return kz_hx_occ<NSR, 3, DUAL>(blocks_per_cu, lds_pad);
    }
    return kz_hx_occ<NSR, 2, DUAL>(blocks_per_cu, lds_pad);
}
template <int NSR, bool DUAL>
static int kz_hx_run_w(int wps, kz_ctx* ctx, const KnnCandParams& p, int n_blocks) {
    if constexpr (NSR <= KZ_HX_WPS3_MAX) {
        if (wps != 2) return kz_hx_run<NSR, 3, DUAL>(ctx, p, n_blocks);
    }
    return kz_hx_run<NSR, 2, DUAL>(ctx, p, n_blocks);
}

#define KZ_DISPATCH_HX(rc, fn, args, DUALV)          \
    do {                                             \
        switch (n_slices) {                          \
            case 2: rc = fn<2, DUALV> args; break;   \
            case 3: rc = fn<3, DUALV> args; break;   \
            case 4: rc = fn<4, DUALV> args; break;   \
            case 5: rc = fn<5, DUALV> args; break;   \
            case 6: rc = fn<6, DUALV> args; break;   \
            case 7: rc = fn<7, DUALV> args; break;   \
            case 8: rc = fn<8, DUALV> args; break;   \
            case 9: rc = fn<9, DUALV> args; break;   \
            case 10: rc = fn<10, DUALV> args; break; \
            case 11: rc = fn<11, DUALV> args; break; \
            case 12: rc = fn<12, DUALV> args; break; \
            case 13: rc = fn<13, DUALV> args; break; \
            case 14: rc = fn<14, DUALV> args; break; \
            case 15: rc = fn<15, DUALV> args; break; \
            case 16: rc = fn<16, DUALV> args; break; \
            case 17: rc = fn<17, DUALV> args; break; \
            case 18: rc = fn<18, DUALV> args; break; \
            case 19: rc = fn<19, DUALV> args; break; \
            case 20: rc = fn<20, DUALV> args; break; \
            case 21: rc = fn<21, DUALV> args; break; \
            case 22: rc = fn<22, DUALV> args; break; \
            case 23: rc = fn<23, DUALV> args; break; \
            case 24: rc = fn<24, DUALV> args; break; \
            default: rc = KZ_ERR_INVALID; break;     \
        }                                            \
    } while (0)

bool kz_hx_supports(int n_slices) { return n_slices >= 2 && n_slices <= 24; }

int kz_hx_occupancy(int n_slices, int dual, int wps, int* blocks_per_cu, int lds_pad) {
    int rc;
    if (dual)
        KZ_DISPATCH_HX(rc, kz_hx_occ_w, (wps, blocks_per_cu, lds_pad), true);
    else
        KZ_DISPATCH_HX(rc, kz_hx_occ_w, (wps, blocks_per_cu, lds_pad), false);
    return rc;
}

int kz_hx_launch(int n_slices, int dual, int wps, kz_ctx* ctx, const KnnCandParams& p, int n_blocks) {
    int rc;
    if (dual)
        KZ_DISPATCH_HX(rc, kz_hx_run_w, (wps, ctx, p, n_blocks), true);
    else
        KZ_DISPATCH_HX(rc, kz_hx_run_w, (wps, ctx, p, n_blocks), false);
    return rc;
}
