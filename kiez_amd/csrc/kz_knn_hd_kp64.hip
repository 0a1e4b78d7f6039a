// dual-pass builds of the fp16 fused kernel, list length 64 (kz_knn_h_inst.h)
#define KZ_H_KP 64
#define KZ_H_DUAL 1
#include "kz_knn_h_inst.h"
