// Small HBM-bound helpers: compact-kernel transposition for backward-data and the fused Adam step.
#include "qk_common.h"

namespace qk {
namespace {

// dst[t][f][p][c] = src[t][c][p][f]  -- backward-data reads the compact kernel with the roles of
// input channel and filter swapped (the transposed convolution of conv.py:334).
__global__ void __launch_bounds__(256)
k_transpose_w(const float *__restrict__ src, float *__restrict__ dst, int taps, int Cq, int F)
{
    __shared__ float tile[32][33];
    const int t = blockIdx.z >> 2, p = blockIdx.z & 3;
    const int c0 = blockIdx.y * 32, f0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, f = f0 + tx;
        tile[r][tx] = (c < Cq && f < F) ? src[((size_t)(t * Cq + c) * 4 + p) * F + f] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int f = f0 + r, c = c0 + tx;
        if (c < Cq && f < F) dst[((size_t)(t * F + f) * 4 + p) * Cq + c] = tile[tx][r];
    }
}

// Keras-2 Adam (keras/optimizers.py Adam.get_updates), the optimiser of working_example.py:106.
__global__ void __launch_bounds__(256)
k_adam(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
       float *__restrict__ v, size_t n, float lr_t, float b1, float b2, float eps, float gscale)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

}  // namespace

int launch_transpose_w(const float *src, float *dst, int taps, int Cq, int F, hipStream_t stream)
{
    dim3 grid((F + 31) / 32, (Cq + 31) / 32, taps * 4);
    hipLaunchKernelGGL(k_transpose_w, grid, dim3(256), 0, stream, src, dst, taps, Cq, F);
    return hipGetLastError() == hipSuccess ? 0 : QK_ERR_LAUNCH;
}

int launch_adam(float *p, const float *g, float *m, float *v, size_t n, float lr, float b1,
                float b2, float eps, int step, float gscale, hipStream_t stream)
{
    if (n == 0) return 0;
    const double t = (double)step;
    const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, stream, p, g, m, v, n, lr_t, b1, b2, eps, gscale);
    return hipGetLastError() == hipSuccess ? 0 : QK_ERR_LAUNCH;
}

}  // namespace qk
