// dual-pass builds of the fp16 fused kernel, list length 16 (kz_knn_h_inst.h)
#define KZ_H_KP 16
#define KZ_H_DUAL 1
#include "kz_knn_h_inst.h"
