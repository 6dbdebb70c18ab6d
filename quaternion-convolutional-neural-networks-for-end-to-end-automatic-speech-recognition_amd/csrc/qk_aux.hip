// Small HBM-bound helpers: the fused Adam step and the tap-folding gather.
#include "qk_common.h"

namespace qk {
namespace {

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

// Tap folding: xcol[m, a*cq2 + t*Cq + c] = x[pos(m,t), a*Cq + c]  (zero in the padding and beyond
// taps*Cq).  One thread per (row, component, group of 8 folded channels); HBM-bound: the output
// (M x 4*cq2) dominates, x itself stays in L2.
template <typename T>
__global__ void __launch_bounds__(256)
k_fold_taps(const T *__restrict__ x, T *__restrict__ xcol, const GemmGeom g, int cq2)
{
    const int groups = cq2 / 8;
    const long long total = (long long)g.M * 4 * groups;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int k8 = (int)(idx % groups);
        const int a = (int)((idx / groups) & 3);
        int m = (int)(idx / (4 * groups));
        const long long dst = (long long)m * 4 * cq2 + a * cq2 + k8 * 8;
        const int o2 = m % g.osp[2]; m /= g.osp[2];
        const int o1 = m % g.osp[1]; m /= g.osp[1];
        const int o0 = m % g.osp[0];
        const int n = m / g.osp[0];
        T v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = k8 * 8 + j;
            const int t = q / g.Q, c = q - t * g.Q;
            float val = 0.f;
            if (t < g.taps) {
                const int t2 = t % g.ks[2];
                const int tt = t / g.ks[2];
                const int t1 = tt % g.ks[1];
                const int t0 = tt / g.ks[1];
                const int i0 = o0 * g.pa[0] + t0 * g.pb[0] + g.pc[0];
                const int i1 = o1 * g.pa[1] + t1 * g.pb[1] + g.pc[1];
                const int i2 = o2 * g.pa[2] + t2 * g.pb[2] + g.pc[2];
                if (i0 >= 0 && i0 < g.isp[0] && i1 >= 0 && i1 < g.isp[1] && i2 >= 0 && i2 < g.isp[2])
                    val = to_f32(x[(long long)n * g.in_sn + (long long)i0 * g.in_ss[0] + (long long)i1 * g.in_ss[1] +
                                   (long long)i2 * g.in_ss[2] + (long long)(a * g.Q + c) * g.in_sc]);
            }
            v[j] = from_f32<T>(val);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) xcol[dst + j] = v[j];
    }
}

}  // namespace

int launch_fold_taps(int dtype, const void *x, void *xcol, const GemmGeom &g, int cq2, hipStream_t stream)
{
    const long long total = (long long)g.M * 4 * (cq2 / 8);
    long long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    switch (dtype) {
    case QK_F32: hipLaunchKernelGGL(k_fold_taps<float>, dim3((unsigned)blocks), dim3(256), 0, stream, (const float *)x, (float *)xcol, g, cq2); break;
    case QK_BF16: hipLaunchKernelGGL(k_fold_taps<bf16>, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16 *)x, (bf16 *)xcol, g, cq2); break;
    case QK_F16: hipLaunchKernelGGL(k_fold_taps<f16>, dim3((unsigned)blocks), dim3(256), 0, stream, (const f16 *)x, (f16 *)xcol, g, cq2); break;
    default: return QK_ERR_INVALID_ARG;
    }
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
