// split-bf16 fused kernels, list length K' = 32 (see kz_knn_bf_inst.h)
#define KZ_BF_KP 32
#include "kz_knn_bf_inst.h"
