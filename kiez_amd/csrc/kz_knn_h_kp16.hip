// fp16 fused kernels, list length K' = 16 (see kz_knn_h_inst.h)
#define KZ_H_KP 16
#include "kz_knn_h_inst.h"
