// Per-list-length translation unit of the split-bf16 kernels (included by kz_knn_bf_kp{16,32,64,128}.hip with KZ_BF_KP
// defined): the slice counts of one list length compile in parallel with the other list lengths.
#include "kz_common.h"
#include "kz_knn_device.h"
#include "kz_knn_bf16.h"

// Largest slice count that still runs the two-workgroups-per-CU kernel (beyond it: one workgroup per CU, overlapped scan)
#ifndef KZ_BF_TWO_WAVE_MAX
#define KZ_BF_TWO_WAVE_MAX 16
#endif
constexpr int KZ_TWM = KZ_BF_TWO_WAVE_MAX;

#define KZ_BF_CAT2(a, b) a##b
#define KZ_BF_CAT(a, b) KZ_BF_CAT2(a, b)

template <int KP, int NSR>
static int kz_bf_occupancy(int* blocks_per_cu, int lds_pad) {
    const void* kern = NSR <= KZ_TWM ? (const void*)kz_knn_cand_bf_kernel<KP, (NSR <= KZ_TWM ? NSR : KZ_TWM), 2>
                                     : (const void*)kz_knn_cand_bf_ov_kernel<KP, (NSR > KZ_TWM ? NSR : KZ_TWM + 1)>;
    const int lds = (NSR <= KZ_TWM ? KZ_BF_LDS : KZ_OV_LDS) + lds_pad;
    KZ_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    int nb = 0;
    KZ_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, lds));
    *blocks_per_cu = nb < 1 ? 1 : nb;
    return KZ_OK;
}

template <int KP, int NSR>
static int kz_launch_bf(kz_ctx* ctx, const KnnCandParams& p, int n_blocks) {
    if (NSR <= KZ_TWM)
        hipLaunchKernelGGL((kz_knn_cand_bf_kernel<KP, (NSR <= KZ_TWM ? NSR : KZ_TWM), 2>), dim3(n_blocks), dim3(256), KZ_BF_LDS + KZ_K_LDS_PAD,
                           ctx->stream, p);
    else
        hipLaunchKernelGGL((kz_knn_cand_bf_ov_kernel<KP, (NSR > KZ_TWM ? NSR : KZ_TWM + 1)>), dim3(n_blocks), dim3(256), KZ_OV_LDS + KZ_K_LDS_PAD,
                           ctx->stream, p);
    KZ_HIP(hipGetLastError());
    return KZ_OK;
}

#define KZ_DISPATCH_BF_NSR(rc, fn, args, KPV)             \
    do {                                                  \
        switch (n_slices_bf) {                            \
            case 2: rc = fn<KPV, 2> args; break;          \
            case 3: rc = fn<KPV, 3> args; break;          \
            case 4: rc = fn<KPV, 4> args; break;          \
            case 5: rc = fn<KPV, 5> args; break;          \
            case 6: rc = fn<KPV, 6> args; break;          \
            case 7: rc = fn<KPV, 7> args; break;          \
            case 8: rc = fn<KPV, 8> args; break;          \
            case 9: rc = fn<KPV, 9> args; break;          \
            case 10: rc = fn<KPV, 10> args; break;        \
            case 11: rc = fn<KPV, 11> args; break;        \
            case 12: rc = fn<KPV, 12> args; break;        \
            case 13: rc = fn<KPV, 13> args; break;        \
            case 14: rc = fn<KPV, 14> args; break;        \
            case 15: rc = fn<KPV, 15> args; break;        \
            case 16: rc = fn<KPV, 16> args; break;        \
            case 17: rc = fn<KPV, 17> args; break;        \
            case 18: rc = fn<KPV, 18> args; break;        \
            case 19: rc = fn<KPV, 19> args; break;        \
            case 20: rc = fn<KPV, 20> args; break;        \
            case 21: rc = fn<KPV, 21> args; break;        \
            case 22: rc = fn<KPV, 22> args; break;        \
            case 23: rc = fn<KPV, 23> args; break;        \
            default: rc = fn<KPV, 24> args; break;        \
        }                                                 \
    } while (0)

int KZ_BF_CAT(kz_bf_occupancy_kp, KZ_BF_KP)(int n_slices_bf, int* blocks_per_cu, int lds_pad) {
    int rc;
    KZ_DISPATCH_BF_NSR(rc, kz_bf_occupancy, (blocks_per_cu, lds_pad), KZ_BF_KP);
    return rc;
}

int KZ_BF_CAT(kz_bf_launch_kp, KZ_BF_KP)(int n_slices_bf, kz_ctx* ctx, const KnnCandParams& p, int n_blocks) {
    int rc;
    KZ_DISPATCH_BF_NSR(rc, kz_launch_bf, (ctx, p, n_blocks), KZ_BF_KP);
    return rc;
}

