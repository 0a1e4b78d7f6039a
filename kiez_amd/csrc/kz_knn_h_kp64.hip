// fp16 fused kernels, list length K' = 64 (see kz_knn_h_inst.h)
#define KZ_H_KP 64
#include "kz_knn_h_inst.h"
