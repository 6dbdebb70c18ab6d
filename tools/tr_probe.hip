// Probe (not product code): semantics of ds_read_b64_tr_b16 on gfx950.
// LDS element value = its 16-bit index; each lane reads 8 bytes at its own address; prints what
// every lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(uint32_t *out, int mode)
{
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned eoff;   // element offset
    if (mode == 0) eoff = 4 * l;                                   // lane-linear, 4 elements per lane
    else if (mode == 1) eoff = (l & 15) * 32 + (l >> 4) * 4;       // 16 rows of 32 elem, 4 col groups
    else eoff = (l >> 4) * 512 + (l & 15) * 4;
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s *)(lds + eoff));
    out[2 * l] = (uint16_t)v[0] | ((uint32_t)(uint16_t)v[1] << 16);
    out[2 * l + 1] = (uint16_t)v[2] | ((uint32_t)(uint16_t)v[3] << 16);
}
int main()
{
    uint32_t *d, h[128];
    hipMalloc(&d, sizeof(h));
    for (int mode = 0; mode < 3; ++mode) {
        k<<<1, 64>>>(d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l)
            printf("lane %2d: %4u %4u %4u %4u%s", l, h[2*l] & 0xffff, h[2*l] >> 16, h[2*l+1] & 0xffff, h[2*l+1] >> 16, (l % 4 == 3) ? "\n" : "   ");
    }
    return 0;
}
