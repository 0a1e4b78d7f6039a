// Device sort used by the dual pass (kz_knn_dual.h): index rows ordered by their event threshold.  rocPRIM's radix sort
// (stable), temporary storage from the context's stream-ordered pool.  Its own translation unit: the rocPRIM
// headers take longer to compile than the rest of the library.
#include "kz_common.h"

#include <rocprim/rocprim.hpp>

int kz_sort_pairs_f32_i32(kz_ctx* ctx, const float* keys_in, float* keys_out, const int* vals_in, int* vals_out, int n, int descending) {
    size_t temp_bytes = 0;
    // (the two directions need the same temporary storage)
    KZ_HIP(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0, 32, ctx->stream));
    void* temp = nullptr;
    if (kz_pool_alloc(ctx, temp_bytes > 0 ? temp_bytes : 16, &temp) != KZ_OK) return KZ_ERR_NOMEM;
    const hipError_t e = descending ? rocprim::radix_sort_pairs_desc(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0, 32, ctx->stream)
                                    : rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0, 32, ctx->stream);
    kz_pool_free(ctx, temp, 0);   // stream-ordered pool: reuse is ordered behind the sort
    if (e != hipSuccess) {
        kz_set_error("kz_sort_pairs_f32_i32: rocprim::radix_sort_pairs failed: %s", hipGetErrorString(e));
        return KZ_ERR_HIP;
    }
    return KZ_OK;
}
