// 16-bit-input MFMA Hamilton implicit GEMM (bf16 / fp16 fast path) -- placeholder until the
// v_mfma_f32_32x32x16_{bf16,f16} kernels land; returning 0 routes the call to the fp32-MFMA path.
#include "qk_common.h"

namespace qk {

int try_hgemm_16(int, const void *, const void *, const float *, const float *, void *, const GemmGeom &,
                 bool, void *, size_t, hipStream_t)
{
    return 0;
}

}  // namespace qk
