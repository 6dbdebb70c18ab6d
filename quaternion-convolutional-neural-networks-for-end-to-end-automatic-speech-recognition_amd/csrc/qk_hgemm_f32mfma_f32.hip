// fp32-MFMA Hamilton implicit GEMM, HBM element type = float (see qk_hgemm_f32mfma.inc)
#include "qk_common.h"
#define QK_T float
#define QK_SUFFIX f32
#define QK_CAT2(a, b) a##b
#define QK_CAT(a, b) QK_CAT2(a, b)
#include "qk_hgemm_f32mfma.inc"
