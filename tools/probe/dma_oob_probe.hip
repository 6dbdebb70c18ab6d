// Probe (not product code): what does an LDS-DMA buffer load (buffer_load_dwordx4 ... lds) write for lanes whose offset is OUT OF RANGE of the buffer resource?
// LDS is pre-filled with 0xdeadbeef; even lanes load in range, odd lanes out of range; the LDS image is printed.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const uint4 *src, uint4 *out, unsigned bytes)
{
    __shared__ __attribute__((aligned(16))) uint4 lds[64];
    lds[threadIdx.x] = make_uint4(0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu);
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(src), 0, (int)bytes, 0x00020000);
    const unsigned voff = (threadIdx.x & 1) ? 0xF0000000u : threadIdx.x * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)lds, 16, (int)voff, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    out[threadIdx.x] = lds[threadIdx.x];
}
int main()
{
    uint4 h[64], *d, *o;
    for (int i = 0; i < 64; ++i) h[i] = make_uint4(i + 1, i + 1, i + 1, i + 1);
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(h));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, sizeof(h));
    hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) printf("lane %d: %08x %08x %08x %08x\n", i, h[i].x, h[i].y, h[i].z, h[i].w);
    int zeros = 0, kept = 0;
    for (int i = 1; i < 64; i += 2) { zeros += h[i].x == 0 && h[i].w == 0; kept += h[i].x == 0xdeadbeefu; }
    printf("out-of-range lanes: %d wrote zeros, %d left LDS untouched (of 32)\n", zeros, kept);
    return 0;
}
