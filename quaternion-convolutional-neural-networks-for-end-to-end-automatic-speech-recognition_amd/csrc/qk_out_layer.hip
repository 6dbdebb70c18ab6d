// The model's output layer, TimeDistributed(Dense(62, activation='softmax')) (interspeech_model.py:171-175 of the reference), as ONE
// hand-written kernel per direction (round 6; until then: a library GEMM with fp32 logits + qk_softmax_rows_* -- the last
// hipBLASLt kernels of the training step):
//
//   forward   y = softmax(x W + b)                          x (rows, K) 16-bit, W (K, U) fp32 master weights, U <= 64
//   backward  dl = y * (dy - <dy, y>)   (the softmax)       dx = dl W^T,   dW += x^T dl,   db += column sums of dl
//
// A stock Keras layer, not a Hamilton product: a plain real GEMM with a 62-wide output.  51 200 rows x 256 -> 62 at B = 256 is
// 1.6 GFLOP for 33 MB forward / 65 MB backward: bandwidth- and latency-bound (bound: hbm), the MFMAs are there to keep the VALU free.
//
//   * v_mfma_f32_16x16x32 with the operands SWAPPED (kernel fragment = A operand, activation fragment = B operand): the accumulator
//     comes out transposed -- a lane holds output row (lane & 15) and the four columns 4 (lane >> 4) .. + 3 of every 16-column
//     tile -- so the softmax of a row is a reduction over a lane's 16 registers plus two cross-lane steps (xor 16, 32), and rows are
//     written in pieces of consecutive columns;
//   * forward: every kernel fragment of the layer (K / 32 x 4 tiles: 128 VGPRs at K = 256) lives in REGISTERS of persistent
//     waves, converted from the fp32 master weights once per workgroup through LDS (no 16-bit copy of W exists anywhere else);
//     the activation fragments are 16-byte global loads (a lane's 8 consecutive K values of its row), the next tile's issued
//     before the current one's MFMAs;
//   * backward: 4-wave workgroups walk 32-row blocks; dl goes to LDS in 16 bits, x through registers into a padded LDS tile;
//     wave w owns x-columns 64 w .. + 63 of both products: dx (K = the 64 padded logit columns; the kernel fragments -- 32 VGPRs --
//     stay in registers; the column order of the fragments is chosen so that a lane ends up with 8 CONSECUTIVE columns: 16-byte
//     stores) and dW (K = the block's 32 rows: both operands are read K-major out of the row-major LDS tiles with the gfx950
//     transposing read ds_read_b64_tr_b16; which rows a K group holds is free -- lo word rows 4 kg .. + 3, hi word rows 16 + 4 kg ..
//     + 3 -- and makes the four rows a half-wave touches fall on different bank groups at pitches = 8 dwords (mod 64) x odd);
//     the 64 x 64 gradient tile of a wave stays in accumulators over all its blocks and goes to the workgroup's slab of the
//     workspace once; a second small kernel sums the slabs in a fixed order (no atomics: bit-repeatable, and 4 M float atomics cost
//     ten times the rest of the kernel).
#include "qk_common.h"

namespace qk {
namespace {

typedef __bf16 o_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 o_f16x8 __attribute__((ext_vector_type(8)));
typedef short o_v4s __attribute__((ext_vector_type(4)));
typedef short o_v8s __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) o_v4s o_lds_v4s;

__device__ __forceinline__ floatx4 mfma_o(bf16, const uint4 &a, const uint4 &b, const floatx4 &c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(o_bf16x8, a), __builtin_bit_cast(o_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ floatx4 mfma_o(f16, const uint4 &a, const uint4 &b, const floatx4 &c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(o_f16x8, a), __builtin_bit_cast(o_f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned pack2o(bf16, float a, float b) { return (unsigned)from_f32<bf16>(a).x | ((unsigned)from_f32<bf16>(b).x << 16); }
__device__ __forceinline__ unsigned pack2o(f16, float a, float b)
{
    const f16 x = (f16)a, y = (f16)b;
    return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
__device__ __forceinline__ float lo16(bf16, unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float hi16(bf16, unsigned v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ float lo16(f16, unsigned v) { return (float)__builtin_bit_cast(f16, (unsigned short)(v & 0xffffu)); }
__device__ __forceinline__ float hi16(f16, unsigned v) { return (float)__builtin_bit_cast(f16, (unsigned short)(v >> 16)); }

typedef unsigned int o_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_o(const void *p, unsigned long long bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)(bytes > 0xffffffffull ? 0xffffffffu : (unsigned)bytes), 0x00020000);
}
__device__ __forceinline__ uint4 ld16_o(__amdgpu_buffer_rsrc_t r, unsigned voff)
{
    const o_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
constexpr unsigned kOob = 0xF0000000u;

// ---- forward -------------------------------------------------------------------------------------------------------------------
// KS = K / 32 (K steps of the 16 x 16 x 32 MFMA).  Wave tile = 16 rows; persistent waves, tile = wave index + i * waves.
template <typename T, int KS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_dense_softmax_fwd(const T *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias, T *__restrict__ y,
                    const long long rows, const int U)
{
    constexpr int K = KS * 32;
    constexpr int WP = K + 8;                        // LDS row pitch in elements: 16 bytes of padding -> the 16 rows of a fragment read sit 4 banks apart
    __shared__ __attribute__((aligned(16))) T wt[64 * WP];          // W^T in 16 bits: [logit column][k]; columns >= U are zero
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t rx = rsrc_o(x, (unsigned long long)rows * K * sizeof(T));
    const long long n_tiles = (rows + 15) / 16;
    const long long stride = (long long)gridDim.x * 4;
    long long tile = (long long)blockIdx.x * 4 + wave;
    uint4 xf[KS], xn[KS];
    auto fetch = [&](long long tl, uint4 (&dst)[KS]) {
        const long long row = tl * 16 + l16;
        const unsigned off = (tl < n_tiles && row < rows) ? (unsigned)((row * K + 8 * g) * (long long)sizeof(T)) : kOob;
#pragma unroll
        for (int s = 0; s < KS; ++s) dst[s] = ld16_o(rx, off == kOob ? kOob : off + 64u * s);
    };
    fetch(tile, xf);                                 // the first tile's rows fly while the kernel is converted
    {
        // thread = (logit column c, K quarter): pairs of consecutive k -> one 4-byte LDS store; 16 pairs of loads in flight at a time
        const int c = tid & 63, kq = tid >> 6;
        constexpr int KPT = K / 4;                   // k values per thread
        const float *wc = w + c;
#pragma unroll
        for (int k0 = 0; k0 < KPT; k0 += 16) {
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = c < U ? wc[(long long)(kq * KPT + k0 + q) * U] : 0.f;
#pragma unroll
            for (int q = 0; q < 16; q += 2) *reinterpret_cast<unsigned *>(wt + c * WP + kq * KPT + k0 + q) = pack2o(T(), v[q], v[q + 1]);
        }
    }
    __syncthreads();
    uint4 wf[4][KS];                                 // A operands: lane = (column 16 t + l16, K group g)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int s = 0; s < KS; ++s) wf[t][s] = *reinterpret_cast<const uint4 *>(wt + (16 * t + l16) * WP + 32 * s + 8 * g);
    // accumulators start at the bias; the padding columns at -1e30: their kernel columns are zero, exp() makes them 0
    float b0[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * t + 4 * g + r;
            b0[t][r] = c < U ? (bias ? bias[c] : 0.f) : -1e30f;
        }
    for (; tile < n_tiles; tile += stride) {
        fetch(tile + stride, xn);                    // (past the end: out-of-range offsets, zeros, nobody reads them)
        floatx4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] = b0[t][r];
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma_o(T(), wf[t][s], xf[s], acc[t]);
        float m = -1e30f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[t][r]);
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc[t][r] = __expf(acc[t][r] - m); sum += acc[t][r]; }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.f / sum;
        const long long row = tile * 16 + l16;
        if (row < rows) {
            T *yr = y + row * U;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int c = 16 * t + 4 * g;
                if ((U & 1) == 0) {                  // even widths: rows are 4-byte aligned, two columns per store
                    if (c + 1 < U) *reinterpret_cast<unsigned *>(yr + c) = pack2o(T(), acc[t][0] * inv, acc[t][1] * inv);
                    if (c + 3 < U) *reinterpret_cast<unsigned *>(yr + c + 2) = pack2o(T(), acc[t][2] * inv, acc[t][3] * inv);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (c + r < U) yr[c + r] = from_f32<T>(acc[t][r] * inv);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) xf[s] = xn[s];
    }
}

// ---- backward ------------------------------------------------------------------------------------------------------------------
// NW = K / 64 waves per workgroup (wave w owns x-columns 64 w .. + 63); block = 32 rows; persistent workgroups.
template <typename T, int NW>
__global__ void __launch_bounds__(NW * 64)
k_dense_softmax_bwd(const T *__restrict__ x, const float *__restrict__ w, const T *__restrict__ y, const T *__restrict__ dy,
                    T *__restrict__ dx, float *__restrict__ part, const long long rows, const int U,
                    const float *__restrict__ dy_scale_dev, const float dy_scale)
{
    constexpr int K = NW * 64, NTHR = NW * 64;
    constexpr int XP = K * 2 + 32;                   // x tile row pitch in bytes: K / 2 + 8 dwords = 8 (mod 64) x odd for K = 64, 128, 256
    constexpr int DP = 160;                          // dl tile row pitch in bytes: 40 dwords = 8 x 5
    static_assert(((XP / 4) % 16) == 8, "x tile pitch: 8 dwords (mod 16) keeps 8 consecutive rows on 8 different 8-bank groups");
    constexpr int RPT = (32 * 8) / NTHR;             // softmax: 8 lanes per row; rows per thread (1 at 256 threads, 2 at 128, 4 at 64)
    constexpr int XU = (32 * K * 2 / 16) / NTHR;     // 16-byte units of the x block per thread (= 4)
    __shared__ __attribute__((aligned(16))) char xs[32 * XP];
    __shared__ __attribute__((aligned(16))) char ds[32 * DP];
    __shared__ float dbs[NTHR / 8][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, g = lane >> 4;

    // kernel fragments of dx = dl W^T for this wave's 64 x-columns (A operands; K = logit column): tile (u, e) holds x-column
    // 64 w + 32 u + 8 (i >> 2) + 4 e + (i & 3) in its row i -- lane (i-group g', register r) of the accumulators of (u, 0) and (u, 1)
    // then holds columns 64 w + 32 u + 8 g' + 0 .. 7: one 16-byte store
    uint4 wf[2][2][2];                               // [u][e][k step]
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int col = 64 * wave + 32 * u + 8 * (l16 >> 2) + 4 * e + (l16 & 3);
                const float *wr = w + (long long)col * U + 32 * s + 8 * g;
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = (32 * s + 8 * g + q < U) ? wr[q] : 0.f;
                wf[u][e][s] = make_uint4(pack2o(T(), v[0], v[1]), pack2o(T(), v[2], v[3]), pack2o(T(), v[4], v[5]), pack2o(T(), v[6], v[7]));
            }
    floatx4 gw[4][4];                                // dW^T tile of this wave: [logit tile t][x-column tile u'], rows = logit columns
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) gw[t][u][r] = 0.f;
    float dbp[8];                      // bias-gradient partials of this thread's 8 columns (pairs p = l8 + 8 q: columns 2p, 2p + 1)
#pragma unroll
    for (int q = 0; q < 8; ++q) dbp[q] = 0.f;

    const __amdgpu_buffer_rsrc_t rx = rsrc_o(x, (unsigned long long)rows * K * sizeof(T));
    // dy enters multiplied by one number: a device scalar (the upstream gradient of a loss node that hands over d loss / d y
    // unscaled -- the mean over the batch of K.ctc_batch_cost, layers._DenseSoftmaxCtcMeanFn) times a host factor (1 / batch x loss scale);
    // d logits is linear in dy: applied to d logits in fp32, in front of its one rounding
    const float sc = (dy_scale_dev ? *dy_scale_dev : 1.f) * dy_scale;
    const int l8 = tid & 7;
    const int np = (U + 1) / 2;                      // column pairs per row (U even on this path: the launcher checks)
    const long long n_blocks = (rows + 31) / 32;
    uint4 xr[XU];
    unsigned yv[RPT][4], gv[RPT][4];
    auto fetch = [&](long long blk) {
#pragma unroll
        for (int k = 0; k < XU; ++k) {
            const int unit = tid + k * NTHR;         // 16-byte unit of the block: row unit / (K / 8), piece unit % (K / 8)
            const long long row = blk * 32 + unit / (K / 8);
            xr[k] = ld16_o(rx, (blk < n_blocks && row < rows) ? (unsigned)(row * K * 2 + (unit % (K / 8)) * 16) : kOob);
        }
#pragma unroll
        for (int rr = 0; rr < RPT; ++rr) {
            const long long row = blk * 32 + (tid >> 3) + rr * (NTHR / 8);
            const bool ok = blk < n_blocks && row < rows;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = l8 + 8 * q;
                const bool okp = ok && p < np;
                yv[rr][q] = okp ? *reinterpret_cast<const unsigned *>(y + row * U + 2 * p) : 0u;
                gv[rr][q] = okp ? *reinterpret_cast<const unsigned *>(dy + row * U + 2 * p) : 0u;
            }
        }
    };
    long long blk = blockIdx.x;
    if (blk < n_blocks) fetch(blk);
    for (; blk < n_blocks; blk += gridDim.x) {
        // ---- softmax backward of the block's rows (8 lanes per row), dl -> LDS in 16 bits; x -> LDS
#pragma unroll
        for (int rr = 0; rr < RPT; ++rr) {
            float dot = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                dot += lo16(T(), yv[rr][q]) * lo16(T(), gv[rr][q]) + hi16(T(), yv[rr][q]) * hi16(T(), gv[rr][q]);
            dot += __shfl_xor(dot, 1);
            dot += __shfl_xor(dot, 2);
            dot += __shfl_xor(dot, 4);
            const int lrow = (tid >> 3) + rr * (NTHR / 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = l8 + 8 * q;
                const unsigned d2 = pack2o(T(), sc * lo16(T(), yv[rr][q]) * (lo16(T(), gv[rr][q]) - dot), sc * hi16(T(), yv[rr][q]) * (hi16(T(), gv[rr][q]) - dot));
                *reinterpret_cast<unsigned *>(ds + lrow * DP + 4 * p) = p < np ? d2 : 0u;       // (pairs 31 .. : the zero padding of the 64-wide tile)
                dbp[2 * q] += lo16(T(), d2);         // the bias gradient is the column sum of what the MFMAs see
                dbp[2 * q + 1] += hi16(T(), d2);
            }
        }
#pragma unroll
        for (int k = 0; k < XU; ++k) {
            const int unit = tid + k * NTHR;
            *reinterpret_cast<uint4 *>(xs + (unit / (K / 8)) * XP + (unit % (K / 8)) * 16) = xr[k];
        }
        __syncthreads();
        fetch(blk + gridDim.x);                      // the next block's loads fly under this block's MFMAs
        // ---- dx[32 rows][64 w .. + 63] = dl (32 x 64) W^T
        {
            floatx4 acc[2][2][2];                    // [row tile][u][e]
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                uint4 dl[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) dl[s] = *reinterpret_cast<const uint4 *>(ds + (16 * rt + l16) * DP + 64 * s + 16 * g);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        floatx4 a = {0.f, 0.f, 0.f, 0.f};
                        a = mfma_o(T(), wf[u][e][0], dl[0], a);
                        a = mfma_o(T(), wf[u][e][1], dl[1], a);
                        acc[rt][u][e] = a;
                    }
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const long long row = blk * 32 + 16 * rt + l16;
                if (row < rows) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const floatx4 a0 = acc[rt][u][0], a1 = acc[rt][u][1];
                        const uint4 v = make_uint4(pack2o(T(), a0[0], a0[1]), pack2o(T(), a0[2], a0[3]), pack2o(T(), a1[0], a1[1]), pack2o(T(), a1[2], a1[3]));
                        *reinterpret_cast<uint4 *>(dx + row * K + 64 * wave + 32 * u + 8 * g) = v;
                    }
                }
            }
        }
        // ---- dW^T[64 logit columns][64 w .. + 63] += dl^T (64 x 32 rows) x (32 rows x 64): K-major fragments by transposing reads.
        // addressing lane L of a 16-lane group: row (lo: 4 g + L / 4, hi: 16 + 4 g + L / 4), columns c0 + 4 (L % 4) .. + 3
        {
            const int tr_row = 4 * g + (l16 >> 2), tr_col = 4 * (l16 & 3);
            uint4 af[4], bf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const char *p = ds + tr_row * DP + (16 * t + tr_col) * 2;
                const o_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((o_lds_v4s *)(p));
                const o_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((o_lds_v4s *)(p + 16 * DP));
                af[t] = __builtin_bit_cast(uint4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const char *p = xs + tr_row * XP + (64 * wave + 16 * u + tr_col) * 2;
                const o_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((o_lds_v4s *)(p));
                const o_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((o_lds_v4s *)(p + 16 * XP));
                bf[u] = __builtin_bit_cast(uint4, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u) gw[t][u] = mfma_o(T(), af[t], bf[u], gw[t][u]);
        }
        __syncthreads();                             // both tiles are free for the next block
    }
    // ---- the workgroup's gradient tile -> its slab of the workspace, registers as they lie (lane-linear 16-byte stores); the bias
    // partials behind it.  k_dense_softmax_reduce sums the slabs in a fixed order.  (Float atomics straight into dW -- 4 M of
    // them from 512 workgroups -- took 236 us of a 258 us launch, measured: the fabric retires ~17 atomics per ns.)
    if (part) {
        float *slab = part + (long long)blockIdx.x * (NW * 16 * 256 + 64);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                *reinterpret_cast<floatx4 *>(slab + ((wave * 16 + t * 4 + u) * 64 + lane) * 4) = gw[t][u];
        // column sums: rows of the block live in tid >> 3, columns in tid & 7 -- fixed-order sum over LDS
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = l8 + 8 * q;
            dbs[tid >> 3][2 * p] = dbp[2 * q];
            dbs[tid >> 3][2 * p + 1] = dbp[2 * q + 1];
        }
        __syncthreads();
        if (tid < 64) {
            float sum = 0.f;
            for (int r = 0; r < NTHR / 8; ++r) sum += dbs[r][tid];
            slab[NW * 16 * 256 + tid] = sum;
        }
    }
}

// dW[(x column) * U + logit column] += sum over slabs, dbias += sum over slabs.  Block = 16 consecutive float4 of the slab layout
// (256 bytes) x 64 groups of slabs: thread (float4 q, group) sums every 64th slab -- all its loads independent, in flight together --
// and the groups meet in LDS; one owner per output element: no atomics, a fixed summation order (bit-repeatable for a given grid).
// (First form: 65 blocks, each thread walking 128 slabs one dependent-looking load at a time: 45 us for 32 MB.)
template <int NW>
__global__ void __launch_bounds__(1024)
k_dense_softmax_reduce(const float *__restrict__ part, const int n_slabs, float *__restrict__ dw, float *__restrict__ dbias, const int U)
{
    __shared__ floatx4 red[64][16];
    constexpr int SLAB = NW * 16 * 256 + 64;
    const int q = threadIdx.x & 15, grp = threadIdx.x >> 4;
    floatx4 sum = {0.f, 0.f, 0.f, 0.f};
    if ((int)blockIdx.x == NW * 64) {                // the last block: the bias gradient (64 floats per slab = 16 float4)
        for (int sl = grp; sl < n_slabs; sl += 64) sum += *reinterpret_cast<const floatx4 *>(part + (long long)sl * SLAB + NW * 16 * 256 + 4 * q);
    } else {
        const float *p = part + ((long long)blockIdx.x * 16 + q) * 4;
#pragma unroll 8
        for (int sl = grp; sl < n_slabs; sl += 64) sum += *reinterpret_cast<const floatx4 *>(p + (long long)sl * SLAB);
    }
    red[grp][q] = sum;
    __syncthreads();
    if (grp < 8) {
        floatx4 v = red[grp][q];
#pragma unroll
        for (int k = 1; k < 8; ++k) v += red[grp + 8 * k][q];
        red[grp][q] = v;
    }
    __syncthreads();
    if (grp != 0) return;
    floatx4 v = red[0][q];
#pragma unroll
    for (int k = 1; k < 8; ++k) v += red[k][q];
    if ((int)blockIdx.x == NW * 64) {
        if (dbias)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * q + r < U) dbias[4 * q + r] += v[r];
        return;
    }
    if (!dw) return;
    const int f = blockIdx.x * 16 + q;               // float4 index of the slab layout: ((wave * 16 + t * 4 + u) * 64 + lane)
    const int reg = f >> 6, lane = f & 63;
    const int wave = reg >> 4, t = (reg >> 2) & 3, u = reg & 3, l16 = lane & 15, g = lane >> 4;
    const int xc = 64 * wave + 16 * u + l16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = 16 * t + 4 * g + r;
        if (c < U) dw[(long long)xc * U + c] += v[r];
    }
}

}  // namespace

bool dense_softmax_supported(int dtype, long long rows, int K, int U)
{
    return (dtype == QK_BF16 || dtype == QK_F16) && rows >= 0 && (K == 64 || K == 128 || K == 256) && U >= 2 && U <= 64 && (U % 2) == 0 &&
           rows * (long long)K * 2 < 0xF0000000ll;
}

int launch_dense_softmax_fwd(int dtype, long long rows, int K, int U, const void *x, const float *w, const float *bias, void *y, hipStream_t stream)
{
    if (!dense_softmax_supported(dtype, rows, K, U)) return QK_ERR_UNSUPPORTED;
    const long long n_tiles = (rows + 15) / 16;
    long long blocks = (long long)device_cu_count() * 2;              // two 4-wave workgroups per CU: 2 waves per SIMD (the kernel fragments take 128 VGPRs)
    if (blocks > (n_tiles + 3) / 4) blocks = (n_tiles + 3) / 4;
    if (blocks < 1) blocks = 1;
#define QK_GO(TT, KS) hipLaunchKernelGGL((k_dense_softmax_fwd<TT, KS>), dim3((unsigned)blocks), dim3(256), 0, stream, (const TT *)x, w, bias, (TT *)y, rows, U)
    if (dtype == QK_BF16) { if (K == 256) QK_GO(bf16, 8); else if (K == 128) QK_GO(bf16, 4); else QK_GO(bf16, 2); }
    else { if (K == 256) QK_GO(f16, 8); else if (K == 128) QK_GO(f16, 4); else QK_GO(f16, 2); }
#undef QK_GO
    return hipGetLastError() == hipSuccess ? 0 : QK_ERR_LAUNCH;
}

// slabs of gradient partials: one per workgroup of the backward kernel (64 x K floats + 64 bias sums each)
static int dense_softmax_bwd_grid(long long rows, int K)
{
    const long long n_blocks = (rows + 31) / 32;
    // workgroups per CU, measured at 51 200 x 256 -> 62 (bf16, us per backward incl. the slab reduction): 1: 33.3, 2: 35.8, 3: 49.6 --
    // every workgroup more is a 64 KB slab written and read again
    long long blocks = (long long)device_cu_count() * (K == 256 ? 1 : K == 128 ? 2 : 4);
    if (blocks > n_blocks) blocks = n_blocks;
    return (int)(blocks < 1 ? 1 : blocks);
}

size_t dense_softmax_bwd_workspace_bytes(int dtype, long long rows, int K, int U)
{
    if (!dense_softmax_supported(dtype, rows, K, U)) return 0;
    return (size_t)dense_softmax_bwd_grid(rows, K) * ((size_t)(K / 64) * 16 * 256 + 64) * sizeof(float);
}

int launch_dense_softmax_bwd(int dtype, long long rows, int K, int U, const void *x, const float *w, const void *y, const void *dy, void *dx,
                             float *dw, float *dbias, const float *dy_scale_dev, float dy_scale, float *ws, hipStream_t stream)
{
    if (!dense_softmax_supported(dtype, rows, K, U)) return QK_ERR_UNSUPPORTED;
    const int blocks = dense_softmax_bwd_grid(rows, K);
    float *part = (dw || dbias) ? ws : nullptr;
#define QK_GO(TT, NW) do { hipLaunchKernelGGL((k_dense_softmax_bwd<TT, NW>), dim3((unsigned)blocks), dim3(NW * 64), 0, stream, (const TT *)x, w, (const TT *)y, (const TT *)dy, (TT *)dx, part, rows, U, dy_scale_dev, dy_scale); \
        if (part) hipLaunchKernelGGL((k_dense_softmax_reduce<NW>), dim3(NW * 64 + 1), dim3(1024), 0, stream, part, blocks, dw, dbias, U); } while (0)
    if (dtype == QK_BF16) { if (K == 256) QK_GO(bf16, 4); else if (K == 128) QK_GO(bf16, 2); else QK_GO(bf16, 1); }
    else { if (K == 256) QK_GO(f16, 4); else if (K == 128) QK_GO(f16, 2); else QK_GO(f16, 1); }
#undef QK_GO
    return hipGetLastError() == hipSuccess ? 0 : QK_ERR_LAUNCH;
}

}  // namespace qk
