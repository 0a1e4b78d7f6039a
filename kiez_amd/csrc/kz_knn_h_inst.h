// Per-list-length translation unit of the fp16 fused kernels (included by kz_knn_h_kp{16,32,64,128}.hip with KZ_H_KP
// defined): the slice counts of one list length compile in parallel with the other list lengths.  With KZ_H_DUAL
// defined (kz_knn_hd_kp*.hip) the unit holds the dual-pass builds of the same kernels (kz_hd_* entry points).
#include "kz_common.h"
#include "kz_knn_device.h"
#include "kz_knn_h16.h"

#ifdef KZ_H_DUAL
#define KZ_H_DUALV true
#define KZ_H_NAME(base) KZ_H_CAT(kz_hd_##base##_kp, KZ_H_KP)
#else
#define KZ_H_DUALV false
#define KZ_H_NAME(base) KZ_H_CAT(kz_h_##base##_kp, KZ_H_KP)
#endif
#define KZ_H_CAT2(a, b) a##b
#define KZ_H_CAT(a, b) KZ_H_CAT2(a, b)

// Occupancy class of a slice count: three workgroups per CU (168 VGPRs, 53 KiB of LDS each) while the stationary query
// tile fits, two (256 VGPRs, 80 KiB) beyond.  wps (tuning knob "h_wps") = 2 forces the two-workgroup build.
// Measured (MI355X, rocprof): d = 128, K' = 16: 2.93 ms at three per CU against 3.51 ms at two; d = 200, K' = 16 (single
// fragment set, kz_knn_h16.h: ONE_SET): 96.8 against 102.2 ms; d = 200, K' = 64 (lists in the output arrays, 13 spilled
// VGPRs at three per CU): 132 against 127 ms -- so beyond 8 slices only the K' = 16 build runs three per CU.
// Tried and dropped: a long-sweep build at three per CU with an 8-slot ring / one barrier per four slices and the lists in
// the output arrays instead of LDS (same-box A/B on ns: 98.0 against 88.7 ms).
constexpr int KZ_H_WPS3_MAX = 8;        // d <= 128: every list length
constexpr int KZ_H_WPS3_MAX_KP16 = 13;  // d <= 208: K' = 16 only

// wide != 0: the wide build (kz_knn_h16.h "WIDE": one workgroup of 4 x WPS waves per CU, WPS query tiles on one ring);
// *tpw = query tiles per workgroup of the kernel returned.
// The wide builds pay where the sweep is long and the epilogue light (K' = 16, more than 8 slices; measured same-box, ms narrow
// -> wide: 250k x 1M x 200: 88.0 -> 85.0, x 300: 124.5 -> 119.3; 100k x 100k x 128: 2.86 -> 2.95; 500k x 500k x 200, K' = 64:
// 103 -> 111: every wave of the CU reaches the tile epilogue at the same time); elsewhere the narrow builds run.
template <int KP, int NSR>
static const void* kz_h_kernel(int wps, int wide_opt, int* lds, int* tpw) {
    constexpr bool three = NSR <= KZ_H_WPS3_MAX || (KP == 16 && NSR <= KZ_H_WPS3_MAX_KP16);
    constexpr bool WIDE_OK = false;   // (round 6: the wide builds are no longer instantiated -- KZ_K_H_WIDE; was KP == 16 && NSR > 8)
    const int wide = WIDE_OK ? wide_opt : 0;
    if (three && wps != 2) {
        constexpr int N3 = three ? NSR : 2;
        if constexpr (WIDE_OK) {
            if (wide) {
                *lds = KzHCfg<KP, 3, N3, KZ_H_DUALV, true>::LDS_BYTES;
                *tpw = 3;
                return (const void*)kz_knn_cand_h_kernel<KP, N3, 3, KZ_H_DUALV, true>;
            }
        }
        *lds = KzHCfg<KP, 3, NSR, KZ_H_DUALV>::LDS_BYTES;
        *tpw = 1;
        return (const void*)kz_knn_cand_h_kernel<KP, N3, 3, KZ_H_DUALV>;
    }
    if constexpr (WIDE_OK) {
        if (wide) {
            *lds = KzHCfg<KP, 2, NSR, KZ_H_DUALV, true>::LDS_BYTES;
            *tpw = 2;
            return (const void*)kz_knn_cand_h_kernel<KP, NSR, 2, KZ_H_DUALV, true>;
        }
    }
    *lds = KzHCfg<KP, 2, NSR, KZ_H_DUALV>::LDS_BYTES;
    *tpw = 1;
    return (const void*)kz_knn_cand_h_kernel<KP, NSR, 2, KZ_H_DUALV>;
}

// *blocks_per_cu = workgroups of the kernel resident per CU, *tpw = query tiles each of them takes
template <int KP, int NSR>
static int kz_h_occupancy(int* blocks_per_cu, int* tpw, int wps, int wide, int lds_pad) {
    int lds = 0;
    const void* kern = kz_h_kernel<KP, NSR>(wps, wide, &lds, tpw);
    KZ_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds + lds_pad));
    int nb = 0;
    KZ_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256 * *tpw, lds + lds_pad));
    *blocks_per_cu = nb < 1 ? 1 : nb;
    return KZ_OK;
}

template <int KP, int NSR>
static int kz_launch_h(kz_ctx* ctx, const KnnCandParams& p, int n_blocks, int wps, int wide) {
    int lds = 0, tpw = 1;
    const void* kern = kz_h_kernel<KP, NSR>(wps, wide, &lds, &tpw);
    KnnCandParams pc = p;
    void* args[] = {&pc};
    KZ_HIP(hipLaunchKernel(kern, dim3(n_blocks), dim3(256 * tpw), args, (size_t)(lds + KZ_K_LDS_PAD), ctx->stream));
    return KZ_OK;
}

#define KZ_DISPATCH_H_NSR(rc, fn, args, KPV)              \
    do {                                                  \
        switch (n_slices) {                               \
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

int KZ_H_NAME(occupancy)(int n_slices, int* blocks_per_cu, int* tpw, int wps, int wide, int lds_pad) {
    int rc;
    KZ_DISPATCH_H_NSR(rc, kz_h_occupancy, (blocks_per_cu, tpw, wps, wide, lds_pad), KZ_H_KP);
    return rc;
}

int KZ_H_NAME(launch)(int n_slices, kz_ctx* ctx, const KnnCandParams& p, int n_blocks, int wps, int wide) {
    int rc;
    KZ_DISPATCH_H_NSR(rc, kz_launch_h, (ctx, p, n_blocks, wps, wide), KZ_H_KP);
    return rc;
}
