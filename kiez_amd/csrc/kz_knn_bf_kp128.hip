// split-bf16 fused kernels, list length K' = 128 (see kz_knn_bf_inst.h)
#define KZ_BF_KP 128
#include "kz_knn_bf_inst.h"
