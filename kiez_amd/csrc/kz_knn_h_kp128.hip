// fp16 fused kernels, list length K' = 128 (see kz_knn_h_inst.h)
#define KZ_H_KP 128
#include "kz_knn_h_inst.h"
