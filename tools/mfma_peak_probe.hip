// Calibration probe (not product code): sustained fp32-MFMA rate and shader clock on this chip.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma(float *out, long long *clk, int iters)
{
    floatx16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-4f;
    long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

int main()
{
    const int blocks = 256, iters = 20000;
    float *out; long long *clk;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k_mfma<4><<<blocks, 256>>>(out, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        double flops = (double)blocks * 4 * iters * 4 * 4096.0;
        printf("NACC=4: %.3f ms  %.1f TFLOP/s   shader cycles %lld (%.1f per MFMA)  wall ticks %lld -> clock %.3f GHz (100 MHz wall)\n",
               ms, flops / ms * 1e-9, h[0], (double)h[0] / (iters * 4.0), h[1], (double)h[0] / h[1] * 0.1);
    }
    return 0;
}
