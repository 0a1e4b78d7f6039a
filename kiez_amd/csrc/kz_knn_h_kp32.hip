// fp16 fused kernels, list length K' = 32 (see kz_knn_h_inst.h)
#define KZ_H_KP 32
#include "kz_knn_h_inst.h"
