// Probe (diagnostic, not product code): does the bf16 MFMA's power depend on WHICH operand carries the sparse data?
// The band kernels multiply kernel fragments (dense values) with activation fragments (65 % exact zeros behind relu + dropout); since round 5
// the kernel fragment is the MFMA's FIRST operand (transposed accumulator).  The chip is power-limited on these kernels
// (DESIGN.md 3.11), so if one operand position gates zeros better than the other, the roles are worth a clock step.
// MFMA-only loop of k_hgemm16_band's wave tile (7 accumulator tiles, fragments fixed in registers), two source images:
//   order 0: mfma(W, X)   (today)        order 1: mfma(X, W)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ floatx16 mfma(const uint4 &a, const uint4 &b, floatx16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
constexpr unsigned kSignConj = 0x284E;
template <int ORDER>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_op(const uint4 *__restrict__ srcw, const uint4 *__restrict__ srcx, float *__restrict__ out, int iters, unsigned units)
{
    const int tid = threadIdx.x;
    floatx16 acc[4], accn[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[b][e] = 0.f; accn[b][e] = 0.f; }
    uint4 X[4], W[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) X[a] = srcx[(blockIdx.x * 977u + tid + a * 256) % units];
#pragma unroll
    for (int p = 0; p < 4; ++p) W[p] = srcw[(blockIdx.x * 613u + tid + p * 256) % units];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if ((kSignConj >> (a * 4 + b)) & 1u) accn[b] = ORDER ? mfma(X[a], W[a ^ b], accn[b]) : mfma(W[a ^ b], X[a], accn[b]);
                    else acc[b] = ORDER ? mfma(X[a], W[a ^ b], acc[b]) : mfma(W[a ^ b], X[a], acc[b]);
                    __builtin_amdgcn_sched_barrier(0);
                }
        // rotate the fragments among themselves so that operand values change from step to step as in a K loop (lane rotation: no memory)
#pragma unroll
        for (int a = 0; a < 4; ++a) { X[a].x = __builtin_amdgcn_mov_dpp(X[a].x, 0x134, 0xf, 0xf, false); }        // wave_shr:1-like shuffle of one dword
    }
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[b][e] - accn[b][e];
    if (s == 123.456f) out[blockIdx.x * 256 + tid] = s;
}
extern "C" double op_launch(int order, int iters, const void *srcw, const void *srcx, unsigned units, float *out, void *stream)
{
    const int grid = 512;
    if (order) hipLaunchKernelGGL((k_op<1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4 *)srcw, (const uint4 *)srcx, out, iters, units);
    else hipLaunchKernelGGL((k_op<0>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4 *)srcw, (const uint4 *)srcx, out, iters, units);
    return (double)grid * 4.0 * iters * 32.0;
}
