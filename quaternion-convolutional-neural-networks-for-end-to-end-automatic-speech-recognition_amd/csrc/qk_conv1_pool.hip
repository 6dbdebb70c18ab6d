// First layer of the TIMIT model as ONE kernel per direction (models/interspeech_model.py:97-103 of the reference):
//
//     QuaternionConv2D(F, (3,5), 'same', relu) on ONE quaternion input channel  ->  MaxPooling2D((1,3), 'same')
//                                                                                   (pools the frequency axis)
//
// The layer moves bytes, not flops (K = 4 x 15, 58 FLOP per byte of its 537 MB output), and as separate launches it moves
// 2.4 GB forward (tap-folded copy of x, y, pooled y) and 1.6 GB backward for 17 MB of input and 184 MB of pooled
// output.  Here the forward reads x and writes ONLY the pooled tensor plus three one-hot bit planes per pooled element
// (window row 0 / 1 / 2 held the maximum and relu let it through; none set = nothing flows back: 38 MB); the backward
// reads x, the pooled gradient and those planes, rebuilds dy in registers and accumulates the kernel / bias gradient
// there -- y never exists in memory.  Both kernels were bound by VALU issue before they were bound by anything else
// (DESIGN.md 3.5.2): the bookkeeping per element is what the code below is organised around.
//
// Geometry: x (N, H, W, 4) channels_last (r,i,j,k of the one channel) or (N, 4, H, W) component planes, kernel
// (KH, KW, 1, 4F), taps KH*KW <= 15; pooled (N, ceil(H / PH), W, 4F).  A persistent workgroup walks over pooled line
// segments (n, ho, 224 positions of W); a wave owns 32 consecutive positions x (4 components x 32 filters) of a segment.
// The x patch (PH + KH - 1 rows x 232 positions x 8 B) sits in LDS, zero padded, the next segment's patch on its way.
//   forward   7 computing waves + a loader wave.  y_b[pos, f] += A_a[pos, tap] * W_{a^b}[tap, f], v_mfma_f32_32x32x16:
//             K = the 16 (15 + 1 zero) taps; lane = filter, registers = positions
//   backward  7 waves.  dW_p[tap, f] += x_a[pos, tap]^T * (+-dy_b)[pos, f], v_mfma_f32_16x16x32: rows = 15 taps + a row of
//             ones (the bias gradient), K = the wave's 32 positions; dy of window row fi = dpool where plane fi is set
// Fragments are assembled from 8-byte LDS reads (all four components of one position) with v_perm_b32.
#include "qk_common.h"
#include <type_traits>

namespace qk {
namespace {

typedef __bf16 c1_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 c1_f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ floatx16 c1_mfma(bf16, const uint4 &a, const uint4 &b, const floatx16 &c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(c1_bf16x8, a), __builtin_bit_cast(c1_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ floatx16 c1_mfma(f16, const uint4 &a, const uint4 &b, const floatx16 &c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(c1_f16x8, a), __builtin_bit_cast(c1_f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ floatx4 c1_mfma16(bf16, const uint4 &a, const uint4 &b, const floatx4 &c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(c1_bf16x8, a), __builtin_bit_cast(c1_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ floatx4 c1_mfma16(f16, const uint4 &a, const uint4 &b, const floatx4 &c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(c1_f16x8, a), __builtin_bit_cast(c1_f16x8, b), c, 0, 0, 0);
}
typedef float c1_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned c1_pack(bf16, float a, float b)
{
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const c1_f2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));
}
__device__ __forceinline__ unsigned c1_pack(f16, float a, float b)
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const c1_f2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2));
}
__device__ __forceinline__ uint4 c1_neg(const uint4 &v)
{
    return make_uint4(v.x ^ 0x80008000u, v.y ^ 0x80008000u, v.z ^ 0x80008000u, v.w ^ 0x80008000u);
}

struct C1Geom {
    int N, H, W, F;            // x (N, H, W, 4); F quaternion filters (output channels 4F)
    int Ho;                    // ceil(H / PH)
    int n_chunks;              // ceil(W / 224)
    int n_lines;               // N * Ho * n_chunks   (work items)
    int has_bias;
    // PRELU variants: slopes (one, or one per row f of the conv output: Keras shared_axes=[1,0] on (C, F, T))
    const float *alpha;
    int alpha_len;
    int x_planes;              // x is (N, 4, H, W): the four component PLANES of the one quaternion channel (channels_first,
                               // how the reference feeds the model: Input(shape=(4, 41, None)), interspeech_model.py:81)
};

constexpr int C1_TW = 224;                 // positions per workgroup (7 waves x 32)
constexpr int C1_PW = C1_TW + 4 + 4;       // patch width in positions (KW - 1 <= 4 halo, + slack for the tap-15 clamp)

// Everything the two kernels share: patch staging, fragment assembly.
template <typename T, int KH, int KW, int PH>
struct C1 {
    static constexpr int TAPS = KH * KW, NR = PH + KH - 1, PADH = (KH - 1) / 2, PADW = (KW - 1) / 2;
    static_assert(TAPS <= 16 && KW - 1 <= 4, "one 16-deep MFMA step holds the taps");
    static constexpr int PATCH_BYTES = NR * C1_PW * 8;

    // x rows [PH*ho - PADH, +NR) x positions [t0 - PADW, +C1_PW) of sample n -> LDS, zeros outside the tensor; in two halves so
    // that a persistent workgroup can have the NEXT segment's loads in flight under this segment's MFMAs (patch_load: global ->
    // registers, nothing consumes them; patch_store: registers -> LDS).
    // x_planes: the r, i, j, k planes are read as they lie -- per plane and patch row a run of consecutive positions,
    // two positions (4 bytes) per lane and plane, i.e. coalesced 256-byte segments per wave and plane, the four loads of a
    // lane in flight together -- and interleaved with v_perm_b32 into the same [position][component] LDS image (two 8-byte
    // stores); odd W or an odd first position falls back to one position per load.
    static constexpr int HALF = C1_PW / 2;
    static_assert(C1_PW % 2 == 0, "patch width in position pairs");
    // NT = the threads that share the staging: the workgroup's 448 (backward), or the 64 of the forward's loader wave
    template <int NT>
    struct Pre {
        static constexpr int IT_CL = (NR * C1_PW + NT - 1) / NT;   // units per thread, channels_last (8 bytes each)
        static constexpr int IT_PL = (NR * HALF + NT - 1) / NT;    // planes (16 bytes each)
        static constexpr int WORDS = 2 * IT_CL > 4 * IT_PL ? 2 * IT_CL : 4 * IT_PL;
        unsigned v[WORDS];
    };
    template <int NT>
    static __device__ __forceinline__ void patch_load(const T *__restrict__ x, const C1Geom &g, int n, int ho, int t0, int tid, Pre<NT> &p)
    {
        const int f_lo = PH * ho - PADH;
        if (g.x_planes) {
            const long long plane = (long long)g.H * g.W;
            if (((g.W | (t0 - PADW)) & 1) == 0) {
                // both positions of a pair share a 4-byte word, and (W, first position even) a pair is inside or outside the tensor
                // as a whole: ONE predicated set of loads per unit -- a per-lane fallback inside this branch would run for every
                // wave that holds a border lane, with its waits (it did: the prefetch was serialised behind vmcnt(0))
#pragma unroll
                for (int it = 0; it < Pre<NT>::IT_PL; ++it) {
                    const int e = tid + it * NT;
                    const int r = e / HALF, c2 = e - r * HALF;
                    const int f = f_lo + r, t = t0 - PADW + 2 * c2;
#pragma unroll
                    for (int a = 0; a < 4; ++a) p.v[4 * it + a] = 0u;       // plane a: positions t (low half), t + 1 (high half)
                    if (e < NR * HALF && f >= 0 && f < g.H && t >= 0 && t < g.W) {
                        const T *row = x + ((long long)n * 4 * g.H + f) * g.W;
#pragma unroll
                        for (int a = 0; a < 4; ++a) p.v[4 * it + a] = *reinterpret_cast<const unsigned *>(row + a * plane + t);   // four loads in flight
                    }
                }
                return;
            }
#pragma unroll
            for (int it = 0; it < Pre<NT>::IT_PL; ++it) {                   // odd W or an odd first position: one position per load
                const int e = tid + it * NT;
                const int r = e / HALF, c2 = e - r * HALF;
                const int f = f_lo + r, t = t0 - PADW + 2 * c2;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    unsigned lo = 0u, hi = 0u;
                    if (e < NR * HALF && f >= 0 && f < g.H) {
                        const T *row = x + ((long long)n * 4 * g.H + f) * g.W;
                        if (t >= 0 && t < g.W) lo = __builtin_bit_cast(unsigned short, row[a * plane + t]);
                        if (t + 1 >= 0 && t + 1 < g.W) hi = __builtin_bit_cast(unsigned short, row[a * plane + t + 1]);
                    }
                    p.v[4 * it + a] = lo | (hi << 16);
                }
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < Pre<NT>::IT_CL; ++it) {
            const int e = tid + it * NT;
            const int r = e / C1_PW, c = e - r * C1_PW;
            const int f = f_lo + r, t = t0 - PADW + c;
            uint2 v = make_uint2(0u, 0u);
            if (e < NR * C1_PW && f >= 0 && f < g.H && t >= 0 && t < g.W)
                v = *reinterpret_cast<const uint2 *>(x + (((long long)n * g.H + f) * g.W + t) * 4);
            p.v[2 * it] = v.x; p.v[2 * it + 1] = v.y;
        }
    }
    template <int NT>
    static __device__ __forceinline__ void patch_store(char *patch, const C1Geom &g, int tid, const Pre<NT> &p)
    {
        if (g.x_planes) {
#pragma unroll
            for (int it = 0; it < Pre<NT>::IT_PL; ++it) {
                const int e = tid + it * NT;
                if (e < NR * HALF) {
                    const unsigned *w4 = p.v + 4 * it;
                    // [position][component] image: two 8-byte stores (r | i, j | k of position t, then of position t + 1)
                    const uint2 p0 = make_uint2(__builtin_amdgcn_perm(w4[1], w4[0], 0x05040100u), __builtin_amdgcn_perm(w4[3], w4[2], 0x05040100u));
                    const uint2 p1 = make_uint2(__builtin_amdgcn_perm(w4[1], w4[0], 0x07060302u), __builtin_amdgcn_perm(w4[3], w4[2], 0x07060302u));
                    *reinterpret_cast<uint2 *>(patch + e * 16) = p0;        // (r * C1_PW + 2 c2) * 8 == e * 16
                    *reinterpret_cast<uint2 *>(patch + e * 16 + 8) = p1;
                }
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < Pre<NT>::IT_CL; ++it) {
            const int e = tid + it * NT;
            if (e < NR * C1_PW) *reinterpret_cast<uint2 *>(patch + e * 8) = make_uint2(p.v[2 * it], p.v[2 * it + 1]);
        }
    }
    // byte offset (relative to the lane's position) of tap k inside the patch; taps >= TAPS are clamped (their B row is zero)
    static __device__ __forceinline__ int tap_off(int k)
    {
        k = k < TAPS ? k : TAPS - 1;
        const int df = k / KW, dt = k - df * KW;
        return (df * C1_PW + dt) * 8;
    }
    // the four per-component fragments of 8 positions-or-taps worth of 8-byte reads: element i of component a
    static __device__ __forceinline__ void split4(const uint2 (&v)[8], uint4 (&A)[4])
    {
        unsigned lo[4], hi[4], lo2[4], hi2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lo[j] = __builtin_amdgcn_perm(v[2 * j + 1].x, v[2 * j].x, 0x05040100u);    // component 0 of elements 2j, 2j+1
            hi[j] = __builtin_amdgcn_perm(v[2 * j + 1].x, v[2 * j].x, 0x07060302u);    // component 1
            lo2[j] = __builtin_amdgcn_perm(v[2 * j + 1].y, v[2 * j].y, 0x05040100u);   // component 2
            hi2[j] = __builtin_amdgcn_perm(v[2 * j + 1].y, v[2 * j].y, 0x07060302u);   // component 3
        }
        A[0] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        A[1] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        A[2] = make_uint4(lo2[0], lo2[1], lo2[2], lo2[3]);
        A[3] = make_uint4(hi2[0], hi2[1], hi2[2], hi2[3]);
    }
    // compact kernel -> the four per-part B fragments of this lane (column = filter j0 + lr, K = taps 8 lh .. 8 lh + 7)
    static __device__ __forceinline__ void load_w(const float *__restrict__ w, int F, int f, int lh, uint4 (&B)[4])
    {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            unsigned d[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k0 = 8 * lh + 2 * j, k1 = k0 + 1;
                const float a = (k0 < TAPS && f < F) ? w[(k0 * 4 + p) * F + f] : 0.f;      // (f >= F: the block's unused filters, round 6 -- F a multiple of 8)
                const float b = (k1 < TAPS && f < F) ? w[(k1 * 4 + p) * F + f] : 0.f;
                d[j] = c1_pack(T(), a, b);
            }
            B[p] = make_uint4(d[0], d[1], d[2], d[3]);
        }
    }
    // A fragments of row tile fi from the lane's 8 tap reads
    static __device__ __forceinline__ void tap_frags(const char *lane_base, const int (&toff)[8], int fi, uint4 (&A)[4])
    {
        uint2 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const uint2 *>(lane_base + fi * (C1_PW * 8) + toff[i]);
        split4(v, A);
    }
};

// ---------------------------------------------------------------------------------------------------------------
// The arg-max side tensor ("aux"): per wave tile and lane six words = three ONE-HOT planes (window row 0 / 1 / 2 held the
// maximum AND the output is alive: relu did not kill it), two words each.  The lane's 64 elements are (component b, accumulator
// register r = 2 j + odd); word h = b >> 1 of a plane holds element (b, 2 j) at bit 8 (1 - (b & 1)) + j and element (b, 2 j + 1)
// sixteen bits above it -- so that the backward masks a PACKED pair of 16-bit gradients with `(word >> k) & 0x00010001` and one
// v_pk_mul_lo_u16.  Stored [tile][plane][lane] as 8-byte words.
//
// The forward builds the planes by SHIFTING comparison results in: v_cmp + v_addc_co_u32 (word = 2 word + carry) is two
// instructions per element and window row, where the read-modify-write of a 2-bit field was four (the kernel is bound by VALU
// issue: 30 VALU instructions per MFMA before).
__device__ __forceinline__ void c1_max_shift_in(unsigned &w, float &m, float a)        // w = 2 w + (a > m);  m = max(m, a)
{
    // all VALU, no lane mask: the SIGN of m - a is the comparison (equal values give +0), v_alignbit_b32 shifts it in.
    // (v_cmp + v_addc_co_u32 + v_cndmask is the same count, but goes through an SGPR pair per element; fmaxf() costs two more
    //  v_max_f32 per call that only quiet signalling NaNs, and so does every builtin that folds to it: the maximum is an asm line.)
    const float d = m - a;
    w = __builtin_amdgcn_alignbit(w, __builtin_bit_cast(unsigned, d), 31);
    // (d is an operand only to ORDER the block behind the subtraction: `a` is an MFMA result, and the wait states that takes are
    //  inserted for instructions the hazard recogniser can see -- a first version that compared inside a block read accumulator
    //  registers 14 / 15 of a tile before the MFMA had written them)
    asm("v_max_f32 %0, %0, %1" : "+v"(m) : "v"(a), "v"(d));
}
__device__ __forceinline__ void c1_shift_in(unsigned &w, bool c) { w = (w << 1) | (c ? 1u : 0u); }
__device__ __forceinline__ unsigned c1_relu_pk(unsigned pk)                              // relu of two packed bf16 / fp16: sign-magnitude, so max
{                                                                                        // with 0 as 16-bit integers
    typedef short s2 __attribute__((ext_vector_type(2)));
    const s2 z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2, pk), z));
}
// raw planes (even / odd registers apart; bit 8 (3 - b) + j) -> the two stored words of one plane
__device__ __forceinline__ uint2 c1_plane_words(unsigned e, unsigned o)
{
    return make_uint2((e >> 16) | (o & 0xFFFF0000u), (e & 0xFFFFu) | (o << 16));
}

// PRELU: the layer is linear + PReLU (interspeech_model.py:97-103 with aact == 'prelu'): the window maximum is taken over
// prelu(conv + bias) with the slope of each row, and the PRE-activation at the arg-max is written beside the pooled
// tensor (`pre_out`): the backward needs it for the derivative and the slope gradient (with the Keras initial slope 0
// it cannot be recovered from the output).
//
// Persistent: a workgroup walks over pooled line segments; the kernel fragments are built once, and the next segment's
// patch is in flight (registers) under this segment's MFMAs, two patch buffers in LDS, one barrier per segment.
template <typename T, int KH, int KW, int PH, bool PRELU>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_conv1_pool_fwd(const T *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias, T *__restrict__ out,
                 T *__restrict__ pre_out, uint2 *__restrict__ argbits, const C1Geom g)
{
    typedef C1<T, KH, KW, PH> K;
    constexpr int EP_PITCH = 80;
    constexpr int PATCH_SLOT = (K::PATCH_BYTES + 15) / 16 * 16;
    __shared__ __attribute__((aligned(16))) char lds[2 * PATCH_SLOT + 7 * 32 * EP_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int j0 = blockIdx.y * 32;
    int item = blockIdx.x;                                        // (n, ho, chunk)
    if (item >= g.n_lines) return;
    if (wave == 7) {
        // The LOADER wave: it alone reads x and fills the patch buffers, one segment ahead of the seven waves that compute.  Its
        // loads never queue behind the workgroup's output stores (one vmcnt counter per wave: with the loads in the computing
        // waves, the wait for the next patch was a wait for the 32 stores issued after them as well).
        typename K::template Pre<64> pre;
#pragma unroll 1
        for (int it = 0; item < g.n_lines; ++it, item += (int)gridDim.x) {
            const int chunk = item % g.n_chunks, line = item / g.n_chunks;
            K::patch_load(x, g, line / g.Ho, line % g.Ho, chunk * C1_TW, lane, pre);
            K::patch_store(lds + (it & 1) * PATCH_SLOT, g, lane, pre);   // (this buffer's readers passed the barrier one segment ago)
            __syncthreads();
        }
        return;
    }
    uint4 B[4];
    K::load_w(w, g.F, j0 + lr, lh, B);
    int toff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) toff[i] = K::tap_off(8 * lh + i);
    float bia4[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) bia4[b] = (g.has_bias && j0 + lr < g.F) ? bias[b * g.F + j0 + lr] : 0.f;
    // The output's way from [filter = lane][positions in registers] to 16-byte runs of filters per position: every lane writes its
    // filter's row of the wave's slab ([filter][32 positions], 64-byte rows, 8 bytes of skew per four rows: the 8-byte writes and
    // the transposing reads both cover all banks) and ds_read_b64_tr_b16 hands a lane four FILTERS of one position -- 4 + 4 LDS
    // instructions per component where 2-byte stores into a [position][filter] slab took 16 + 2.
    char *ep = lds + 2 * PATCH_SLOT + wave * (32 * EP_PITCH);
    const int ep_w = lr * 64 + (lr >> 2) * 8 + lh * 8;            // + 16 q: positions 4 lh + 8 q .. + 3
    const int ep_fr = 8 * (lane & 3) + ((lane & 15) >> 2);        // the filter row this lane ADDRESSES in a transposing read (it receives filters 8 (i >> 2) + 0 .. 3)
    const int ep_r = ep_fr * 64 + (ep_fr >> 2) * 8 + (lane >> 4) * 8;        // + 32 pass: positions 16 pass + 4 (lane >> 4) .. + 3
    const int e_row = 4 * (lane >> 4) + (lane & 3), e_chunk = (lane & 15) >> 2;      // what it receives: position e_row (+ 16 pass), filters 8 e_chunk .. + 7
    // Round 6: F need not fill the 32-filter block (start_filter = 16 models): a lane whose 8 filters lie past F sends its stores past the
    // line's buffer bound, where they are dropped like the rows past W (was: kernel zero-padded to 32 filters + slice / pad passes in torch)
    const bool e_live = j0 + 8 * e_chunk < g.F;

#pragma unroll 1
    for (int it = 0; item < g.n_lines; ++it, item += (int)gridDim.x) {
        __syncthreads();                                          // patch `it` is in its buffer
        const int chunk = item % g.n_chunks, line = item / g.n_chunks;
        const int ho = line % g.Ho, n = line / g.Ho;
        const int t0 = chunk * C1_TW;
        const int tw = t0 + wave * 32;                            // first position of this wave's tile
        if (tw < g.W) {
            const char *lane_base = lds + (it & 1) * PATCH_SLOT + (wave * 32 + lr) * 8;
            // window maximum and WHICH row tile holds it (first maximum wins, as in TF / torch), one component b at a time: the
            // fragments of the window's rows are built once, the maxima of one component live in 16 registers
            const int n_fi = min(PH, g.H - PH * ho);              // partial last window ('same' pooling: high side only)
            uint4 A[PH][4];
            float af[PH];
#pragma unroll
            for (int fi = 0; fi < PH; ++fi) {
                K::tap_frags(lane_base, toff, fi < n_fi ? fi : 0, A[fi]);
                af[fi] = PRELU ? g.alpha[g.alpha_len > 1 ? min(PH * ho + fi, g.alpha_len - 1) : 0] : 0.f;
            }
            unsigned ue[PH], uo[PH];                              // [fi >= 1]: "row fi beat the rows before it", shifted in b ascending, r descending
#pragma unroll
            for (int fi = 0; fi < PH; ++fi) { ue[fi] = 0u; uo[fi] = 0u; }
            uint2 alive = PRELU ? make_uint2(~0u, ~0u) : make_uint2(0u, 0u);                    // in the stored layout
            T *line_out = out + (((long long)n * g.Ho + ho) * g.W) * (4 * g.F) + j0;           // (wave-uniform: 32-bit lane offsets below)
            T *line_pre = PRELU && pre_out ? pre_out + (((long long)n * g.Ho + ho) * g.W) * (4 * g.F) + j0 : nullptr;
            // rows past W fall outside the line's buffer resource: their stores are dropped, no branch splits the schedule below
            const unsigned line_bytes = (unsigned)((g.W * 4 * g.F - j0) * 2);
            const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(line_out, 0, (int)line_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t pre_rs = __builtin_amdgcn_make_buffer_rsrc(PRELU && line_pre ? line_pre : line_out, 0, (int)line_bytes, 0x00020000);
            // behind the MFMAs of one component b, first half: window rows 0 and 1
            auto fin1 = [&](int b, const floatx16 &a0, const floatx16 &a1, floatx16 &pooled, floatx16 &presel, bool has1) {
                pooled = a0;
                if constexpr (PRELU) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        presel[r] = pooled[r] + bia4[b];
                        pooled[r] = fmaxf(presel[r], 0.f) + af[0] * fminf(presel[r], 0.f);
                    }
                }
                if (!has1) return;
#pragma unroll
                for (int r = 15; r >= 0; --r) {
                    unsigned &u = (r & 1) ? uo[1] : ue[1];
                    if constexpr (PRELU) {
                        const float pv = a1[r] + bia4[b];
                        const float act = fmaxf(pv, 0.f) + af[1] * fminf(pv, 0.f);
                        const bool up = act > pooled[r];
                        pooled[r] = up ? act : pooled[r];
                        presel[r] = up ? pv : presel[r];
                        c1_shift_in(u, up);
                    } else {
                        float m = pooled[r];
                        c1_max_shift_in(u, m, a1[r]);
                        pooled[r] = m;
                    }
                }
            };
            // second half: window row 2, relu / alive, transpose through LDS, 16-byte stores
            auto fin2 = [&](int b, const floatx16 &a2, floatx16 &pooled, floatx16 &presel, bool has2) {
                if (has2) {
#pragma unroll
                    for (int r = 15; r >= 0; --r) {
                        unsigned &u = (r & 1) ? uo[2] : ue[2];
                        if constexpr (PRELU) {
                            const float pv = a2[r] + bia4[b];
                            const float act = fmaxf(pv, 0.f) + af[2] * fminf(pv, 0.f);
                            const bool up = act > pooled[r];
                            pooled[r] = up ? act : pooled[r];
                            presel[r] = up ? pv : presel[r];
                            c1_shift_in(u, up);
                        } else {
                            float m = pooled[r];
                            c1_max_shift_in(u, m, a2[r]);
                            pooled[r] = m;
                        }
                    }
                }
                // relu(max + bias) (== max of relu(conv + bias)) on the PACKED 16-bit pairs, and which outputs are alive: min(value, 1)
                // as 16-bit integers is the pair's two alive bits, sixteen bits apart -- shifted in as they will be stored
                unsigned pk[8];
                if constexpr (!PRELU) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) pooled[r] += bia4[b];
#pragma unroll
                    for (int j = 7; j >= 0; --j) {
                        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                        pk[j] = c1_relu_pk(c1_pack(T(), pooled[2 * j], pooled[2 * j + 1]));
                        const us2 one = {1, 1};
                        const unsigned t = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(us2, pk[j]), one));
                        unsigned &ax = (b >> 1) ? alive.y : alive.x;
                        ax = (ax << 1) | t;
                    }
                }
#pragma unroll
                for (int which = 0; which < (PRELU ? 2 : 1); ++which) {
                    if (which && !line_pre) break;
                    if constexpr (PRELU) {
#pragma unroll
                        for (int r = 0; r < 16; r += 2) pk[r >> 1] = which ? c1_pack(T(), presel[r], presel[r + 1]) : c1_pack(T(), pooled[r], pooled[r + 1]);
                    }
                    // registers 4 q .. 4 q + 3 are four consecutive positions of this lane's filter: one 8-byte word of its row
#pragma unroll
                    for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2 *>(ep + ep_w + q * 16) = make_uint2(pk[2 * q], pk[2 * q + 1]);
#pragma unroll
                    for (int pass = 0; pass < 2; ++pass) {
                        typedef short v4s __attribute__((ext_vector_type(4)));
                        typedef __attribute__((address_space(3))) v4s lds_v4s;
                        const v4s f03 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(ep + ep_r + pass * 32));
                        const v4s f47 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(ep + ep_r + pass * 32 + 4 * 64 + 8));
                        const uint2 lo = __builtin_bit_cast(uint2, f03), hi = __builtin_bit_cast(uint2, f47);
                        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                        const u32x4_t vv = {lo.x, lo.y, hi.x, hi.y};
                        __builtin_amdgcn_raw_buffer_store_b128(vv, which ? pre_rs : out_rs,
                                                               e_live ? (int)(((tw + e_row + 16 * pass) * (4 * g.F) + b * g.F + e_chunk * 8) * 2) : 0x7ffffff0, 0, 0);
                    }
                }
            };
            // MFMAs of component b for window rows [f0, f1), round robin (a dependent MFMA is f1 - f0 issues behind its source)
            auto conv_rows = [&](int b, int f0, int f1, floatx16 (&acc)[PH]) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const bool ng = (kSignConv >> (a * 4 + b)) & 1u;
                    uint4 bw = B[a ^ b];
                    if (ng) bw = c1_neg(bw);                      // (four VALU for up to three MFMAs: cheaper than 16 registers of negated fragments)
#pragma unroll
                    for (int fi = f0; fi < f1; ++fi) {
                        if (a == 0) {
                            floatx16 z;
#pragma unroll
                            for (int r = 0; r < 16; ++r) z[r] = 0.f;
                            acc[fi] = c1_mfma(T(), A[fi][a], bw, z);
                        } else acc[fi] = c1_mfma(T(), A[fi][a], bw, acc[fi]);
                    }
                }
            };
            if (n_fi == PH && !PRELU) {
                // Full windows.  A wave issues in order, and four dependent MFMAs back to back hold it for ~130 cycles: the MFMAs
                // are therefore ISSUED between the VALU instructions of the step before them --
                //   rows 0, 1 of component b + 1 (8 MFMAs)  under  row 2, relu, planes, transpose of component b   (~100 VALU)
                //   row 2 of component b         (4 MFMAs)  under  rows 0, 1 of component b                         (~50 VALU)
                floatx16 acc[PH], nacc[PH], pooled, presel;
                conv_rows(0, 0, 2, acc);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    conv_rows(b, 2, 3, acc);
                    fin1(b, acc[0], acc[1], pooled, presel, true);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);         // twelve VALU
                    }
                    if (b + 1 < 4) conv_rows(b + 1, 0, 2, nacc);
                    fin2(b, acc[2], pooled, presel, true);
                    if (b + 1 < 4) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
                        }
                        acc[0] = nacc[0]; acc[1] = nacc[1];
                    }
                }
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    floatx16 acc[PH], pooled, presel;
                    if constexpr (PRELU) __builtin_amdgcn_sched_barrier(0);
                    conv_rows(b, 0, PH, acc);
                    fin1(b, acc[0], acc[1], pooled, presel, n_fi > 1);
                    fin2(b, acc[2], pooled, presel, n_fi > 2);
                }
            }
            if (argbits) {
                static_assert(PH == 3, "three one-hot planes");
                uint2 *ab = argbits + ((((long long)blockIdx.y * g.n_lines + item) * 7 + wave) * 3) * 64 + lane;
                const uint2 u1 = c1_plane_words(ue[1], uo[1]), u2 = c1_plane_words(ue[2], uo[2]);
                ab[0] = make_uint2(alive.x & ~(u1.x | u2.x), alive.y & ~(u1.y | u2.y));
                ab[64] = make_uint2(alive.x & u1.x & ~u2.x, alive.y & u1.y & ~u2.y);
                ab[128] = make_uint2(alive.x & u2.x, alive.y & u2.y);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// PRELU: dy of the arg-max row = dpool * prelu'(pre) with that row's slope; d alpha[row] += dpool * min(pre, 0).  Both are
// applied while the pooled gradient is STAGED (the thread that moves a 16-byte unit holds dpool and the pre-activation
// `pre_sel` of the same 8 elements; which row won comes from the wave's arg-max words, parked in LDS first), so the MFMA
// part of the loop is the relu variant's, unchanged.
template <typename T, int KH, int KW, int PH, bool PRELU>
__global__ void __launch_bounds__(448) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_conv1_pool_bwd(const T *__restrict__ x, const T *__restrict__ dout, const T *__restrict__ pre_sel, const uint2 *__restrict__ argbits,
                 float *__restrict__ dw, float *__restrict__ dbias, float *__restrict__ dalpha, const C1Geom g)
{
    typedef C1<T, KH, KW, PH> K;
    // a wave's dpool tile: [component 4][position 32][filter 32] 16-bit values, rows of 64 bytes (four rows = all 64 banks: the
    // transposing reads below are conflict free), components 64 bytes askew (so are the 16-byte staging stores)
    constexpr int DP_COMP = 32 * 64 + 64, DP_WAVE = 4 * DP_COMP;
    constexpr int DP_BYTES = 7 * DP_WAVE;
    constexpr int ONES_BYTES = C1_PW * 8;                         // a patch row of (1, 0, 0, 0): the bias gradient's "tap" (below)
    constexpr int PATCH_SLOT = (K::PATCH_BYTES + 15) / 16 * 16;
    constexpr int AW_BYTES = PRELU ? 7 * 64 * 24 : 0;             // the waves' arg-max planes
    // two patch buffers: ONE barrier per segment (the dpool tiles are private to their waves: program order is enough for them)
    __shared__ __attribute__((aligned(16))) char lds[2 * PATCH_SLOT + ONES_BYTES + DP_BYTES + AW_BYTES + (PRELU ? 256 : 0)];
    constexpr int ONES_AT = 2 * PATCH_SLOT, DP_AT = ONES_AT + ONES_BYTES, AW_AT = DP_AT + DP_BYTES;
    float *dal_s = reinterpret_cast<float *>(lds + AW_AT + AW_BYTES);                           // PRELU: 64 slope-gradient sums
    if (PRELU && threadIdx.x < 64) dal_s[threadIdx.x] = 0.f;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int j0 = blockIdx.y * 32;
    for (int e = tid; e < C1_PW; e += 448)
        *reinterpret_cast<uint2 *>(lds + ONES_AT + e * 8) = make_uint2(c1_pack(T(), 1.f, 0.f), 0u);
    // The gradient GEMM runs on v_mfma_f32_16x16x32: dW_p[tap 16, filter 16 h ..] += x_a^T [tap, 32 positions] * (+-dy_b)[32 positions,
    // filter] -- the 15 taps + one bias row fill the 16 rows (the 32 x 32 x 16 form spent half of its MFMAs on 16 rows nobody read).
    //   lane L: row / column L & 15, K group kg = L >> 4: eight of the wave's 32 positions.  WHICH eight is free as long as both operands
    //   agree; they are the positions of accumulator registers 8 (kg & 1) .. + 7 of the forward's lanes lh = kg >> 1 -- position
    //   4 (kg >> 1) + 16 (kg & 1) + (i & 3) + 8 (i >> 2) for K slot i -- because that is the order the arg-max planes are in: this
    //   lane uses the plane words of forward lane (filter, kg >> 1), shifted right by 4 (kg & 1) pairs.
    // A operand (x^T): this lane's row = tap L & 15 (rows 0 - 14).  Row 15 reads a row of ones instead (component 0 only): accumulator
    // row 15 of part b is then sum over positions of dy_b -- the BIAS gradient comes out of the same MFMAs.
    const int kg = lane >> 4, l16 = lane & 15;
    const int my_tap_off = l16 < 15 ? K::tap_off(l16) : ONES_AT;
    const int my_buf_pitch = l16 < 15 ? PATCH_SLOT : 0;
    const int my_fi_pitch = l16 < 15 ? C1_PW * 8 : 0;
    const int my_pos0 = 4 * (kg >> 1) + 16 * (kg & 1);            // first of this lane's eight positions
    floatx4 dwacc[4][2];                                          // [part p][filter half h]: rows = taps 4 kg + r, columns = filters 16 h + (L & 15)
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) dwacc[p][h][r] = 0.f;
    // transposing read (ds_read_b64_tr_b16): lane l of a 16-lane group addresses row l >> 2, filters 4 (l & 3) .. + 3 of the
    // group's 16 filters and receives ITS filter at the group's four rows: a lane's dy fragment is two such reads (its positions
    // + 0..3 and + 8..11) per filter half
    // Banks: the two K groups of one 32-lane pass read rows 16 apart -- the same banks in 64-byte rows -- so rows 16 - 31 are stored
    // with their 32-byte halves SWAPPED: a pass then covers all 64 banks once (4.1e6 -> conflict cycles without it, SQ_LDS_BANK_CONFLICT).
    const int tr_swap = 32 * (kg & 1);                            // my_pos0 & 16
    const int tr_off = DP_AT + wave * DP_WAVE + (my_pos0 + (l16 >> 2)) * 64 + (4 * (lane & 3)) * 2;

    // Persistent: a workgroup walks over pooled line segments and keeps the gradients in registers (one flush of 2 K
    // atomics per workgroup instead of per segment).  The relu form has the NEXT segment's x patch, pooled gradient and planes
    // in flight (50 registers) under this segment's MFMAs; the PReLU form (which turns dpool into dy while staging, with the
    // pre-activations beside it) loads and stores back to back.
    typename K::template Pre<448> pre_patch;
    uint4 pre_dp[8];
    uint2 pre_X[3][2];                                            // [plane][filter half]
    auto load_item = [&](int item) {
        const int chunk = item % g.n_chunks, line = item / g.n_chunks;
        const int ho = line % g.Ho, n = line / g.Ho;
        const int t0 = chunk * C1_TW, tw = t0 + wave * 32;
        K::patch_load(x, g, n, ho, t0, tid, pre_patch);
#pragma unroll
        for (int fi = 0; fi < 3; ++fi) pre_X[fi][0] = pre_X[fi][1] = make_uint2(0u, 0u);       // no plane set: nothing flows
        if (tw < g.W) {
            const uint2 *ab = argbits + ((((long long)blockIdx.y * g.n_lines + item) * 7 + wave) * 3) * 64 + 32 * (kg >> 1) + l16;
#pragma unroll
            for (int fi = 0; fi < 3; ++fi)
#pragma unroll
                for (int h = 0; h < 2; ++h) pre_X[fi][h] = ab[fi * 64 + 16 * h];
        }
        const T *line_in = dout + (((long long)n * g.Ho + ho) * g.W) * (4 * g.F) + j0;          // (wave-uniform)
#pragma unroll
        for (int u0 = 0; u0 < 8; ++u0) {                           // 16-byte units: row u / 16, (component, 8 filters) u % 16
            const int u = lane + 64 * u0;
            const int row = u >> 4, q = u & 15, b = q >> 2, sub = (q & 3) * 8;
            pre_dp[u0] = make_uint4(0u, 0u, 0u, 0u);               // rows past W are zero
            if (tw + row < g.W && j0 + sub < g.F) pre_dp[u0] = *reinterpret_cast<const uint4 *>(line_in + (unsigned)((tw + row) * (4 * g.F) + b * g.F + sub));
        }
    };
    if (!PRELU && (int)blockIdx.x < g.n_lines) load_item(blockIdx.x);
    if constexpr (PRELU) __syncthreads();                          // the slope-gradient sums are zero before the first segment adds to them
    int it = 0;
#pragma unroll 1
    for (int item0 = blockIdx.x; item0 < g.n_lines; item0 += gridDim.x, it ^= 1) {
        int item = item0;
        asm volatile("" : "+s"(item));
        const int chunk = item % g.n_chunks, line = item / g.n_chunks;
        const int ho = line % g.Ho, n = line / g.Ho;
        const int t0 = chunk * C1_TW;
        // this wave's dpool tile: 32 positions x (4 x 32) channels
        char *dp = lds + DP_AT + wave * DP_WAVE;
        const int tw = t0 + wave * 32;
        // (no barrier here: this patch buffer was last read two segments ago, before the previous segment's barrier)
        if constexpr (PRELU) load_item(item);
        K::patch_store(lds + it * PATCH_SLOT, g, tid, pre_patch);
        unsigned X[3][2][2];                                      // [plane][filter half][word], shifted to this lane's registers
#pragma unroll
        for (int fi = 0; fi < 3; ++fi)
#pragma unroll
            for (int h = 0; h < 2; ++h) { X[fi][h][0] = pre_X[fi][h].x >> (4 * (kg & 1)); X[fi][h][1] = pre_X[fi][h].y >> (4 * (kg & 1)); }
        float a3[3] = {0.f, 0.f, 0.f}, dal3[3] = {0.f, 0.f, 0.f};
        if constexpr (PRELU) {
            // the planes in the FORWARD's lane order, parked for the staging threads
            // (a wave's own LDS accesses execute in order: the reads below see these writes without a barrier)
            unsigned *awp = reinterpret_cast<unsigned *>(lds + AW_AT + (wave * 64 + lane) * 24);
            uint2 own[3] = {make_uint2(0u, 0u), make_uint2(0u, 0u), make_uint2(0u, 0u)};
            if (tw < g.W) {
                const uint2 *ab = argbits + ((((long long)blockIdx.y * g.n_lines + item) * 7 + wave) * 3) * 64 + lane;
                own[0] = ab[0]; own[1] = ab[64]; own[2] = ab[128];
            }
            awp[0] = own[0].x; awp[1] = own[0].y; awp[2] = own[1].x; awp[3] = own[1].y; awp[4] = own[2].x; awp[5] = own[2].y;
#pragma unroll
            for (int fi = 0; fi < 3; ++fi) a3[fi] = g.alpha[g.alpha_len > 1 ? min(PH * ho + fi, g.alpha_len - 1) : 0];
        }
        const T *line_pre = PRELU ? pre_sel + (((long long)n * g.Ho + ho) * g.W) * (4 * g.F) + j0 : nullptr;
#pragma unroll
        for (int u0 = 0; u0 < 8; ++u0) {
            const int u = lane + 64 * u0;
            const int row = u >> 4, q = u & 15, b = q >> 2, sub = (q & 3) * 8;
            uint4 v = pre_dp[u0];
            if constexpr (PRELU) {
                if (tw + row < g.W && j0 + sub < g.F) {
                    const uint4 pv = *reinterpret_cast<const uint4 *>(line_pre + (unsigned)((tw + row) * (4 * g.F) + b * g.F + sub));
                    // element (row, filter sub + e) sits in lane (sub + e) + 32 lh', register r' of the accumulator layout
                    const int rr = (row & 3) + 4 * (row >> 3);
                    const int bit = 8 * (1 - (b & 1)) + (rr >> 1) + 16 * (rr & 1);
                    const char *awb = lds + AW_AT + (wave * 64 + 32 * ((row >> 2) & 1) + sub) * 24 + (b >> 1) * 4;
                    unsigned dv[4] = {v.x, v.y, v.z, v.w}, pw[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        float gq[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const unsigned *pl = reinterpret_cast<const unsigned *>(awb + (e + h) * 24);
                            const bool is0 = (pl[0] >> bit) & 1u, is1 = (pl[2] >> bit) & 1u, is2 = (pl[4] >> bit) & 1u;
                            const unsigned short db_ = (unsigned short)(dv[e >> 1] >> (16 * h)), pb_ = (unsigned short)(pw[e >> 1] >> (16 * h));
                            const float d = to_f32(__builtin_bit_cast(T, db_)), pr = to_f32(__builtin_bit_cast(T, pb_));
                            const float al = is0 ? a3[0] : is1 ? a3[1] : a3[2];
                            const float hneg = d * fminf(pr, 0.f);
                            dal3[0] += is0 ? hneg : 0.f; dal3[1] += is1 ? hneg : 0.f; dal3[2] += is2 ? hneg : 0.f;
                            gq[h] = d * (pr > 0.f ? 1.f : (pr < 0.f ? al : 0.f));
                        }
                        dv[e >> 1] = c1_pack(T(), gq[0], gq[1]);
                    }
                    v = make_uint4(dv[0], dv[1], dv[2], dv[3]);
                }
            }
            *reinterpret_cast<uint4 *>(dp + b * DP_COMP + row * 64 + ((sub * 2 + 2 * (row & 16)) & 63)) = v;     // (rows 16 - 31: halves swapped, see tr_off)
        }
        if constexpr (PRELU) {
            if (dalpha) {                                          // wave sums -> one LDS atomic per (wave, window row)
#pragma unroll
                for (int fi = 0; fi < 3; ++fi) {
                    float t = dal3[fi];
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
                    if (lane == 0 && PH * ho + fi < g.H) atomicAdd(dal_s + (g.alpha_len > 1 ? PH * ho + fi : 0), t);
                }
            }
        }
        __syncthreads();
        if constexpr (!PRELU) {
            if (item0 + (int)gridDim.x < g.n_lines) load_item(item0 + (int)gridDim.x);
        }
        const int n_fi = tw < g.W ? min(PH, g.H - PH * ho) : 0;
        // the pooled gradient of this lane's two filters at its eight positions: D[b][h], read ONCE for the three window rows
        uint4 D[4][2];
        if (n_fi > 0) {
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    typedef short v4s __attribute__((ext_vector_type(4)));
                    typedef __attribute__((address_space(3))) v4s lds_v4s;
                    const char *src = lds + tr_off + b * DP_COMP + ((h * 32) ^ tr_swap);
                    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(src));
                    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(src + 8 * 64));
                    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    D[b][h] = make_uint4(l2.x, l2.y, h2.x, h2.y);
                }
        }
#pragma unroll 1
        for (int fi = 0; fi < n_fi; ++fi) {                        // dy of row tile fi (where fi is the window's arg-max), dW += x^T dy
            // x^T fragments: row = this lane's tap, K slot i = position my_pos0 + (i & 3) + 8 (i >> 2)
            uint4 XT[4];
            {
                uint2 v[8];
                const char *xb = lds + it * my_buf_pitch + (wave * 32 + my_pos0) * 8 + fi * my_fi_pitch + my_tap_off;
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const uint2 *>(xb + ((i & 3) + 8 * (i >> 2)) * 8);
                K::split4(v, XT);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned xw = fi == 0 ? X[0][h][b >> 1] : fi == 1 ? X[1][h][b >> 1] : X[2][h][b >> 1];
                    const unsigned dd[4] = {D[b][h].x, D[b][h].y, D[b][h].z, D[b][h].w};
                    unsigned d[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {                  // K slots 2 i, 2 i + 1 = registers 8 (kg & 1) + 2 i, + 1 of the forward
                        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                        const unsigned m = (xw >> (8 * (1 - (b & 1)) + i)) & 0x00010001u;
                        d[i] = __builtin_bit_cast(unsigned, __builtin_bit_cast(us2, dd[i]) * __builtin_bit_cast(us2, m));
                    }
                    const uint4 dy = make_uint4(d[0], d[1], d[2], d[3]);
                    const uint4 dyn = c1_neg(dy);
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const bool ng = (kSignConv >> (a * 4 + b)) & 1u;
                        dwacc[a ^ b][h] = c1_mfma16(T(), XT[a], ng ? dyn : dy, dwacc[a ^ b][h]);
                    }
                }
            }
        }
    }
    if constexpr (PRELU) {
        if (dalpha) {
            __syncthreads();
            if (tid < g.alpha_len && tid < 64 && dal_s[tid] != 0.f) atomicAdd(dalpha + tid, dal_s[tid]);
        }
    }
    // ---- flush: every wave parks its gradients in its own LDS slab (accumulator register r of lane L = tap 4 kg + r, filter
    // 16 h + (L & 15); the bias sums are tap 15), the workgroup sums the seven slabs and issues one atomic per element
    __syncthreads();
    constexpr int SLAB = 4 * 16 * 32 + 4 * 64;                    // [part 4][tap 16][filter 32] + [component 4][filter 32 (+ 32 unused)] floats
    static_assert(7 * SLAB * 4 <= (int)sizeof(lds), "flush slabs fit");
    float *slab = reinterpret_cast<float *>(lds) + wave * SLAB;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[(p * 16 + 4 * kg + r) * 32 + 16 * h + l16] = dwacc[p][h][r];
            if (kg == 3) slab[4 * 16 * 32 + p * 64 + 16 * h + l16] = dwacc[p][h][3];            // (a = 0 carries no sign: kSignConv bits 0 - 3)
        }
    static_assert((kSignConv & 0xFu) == 0u, "component 0 of x enters every part with +");
    __syncthreads();
    const float *all = reinterpret_cast<const float *>(lds);
    for (int e = tid; e < 4 * 16 * 32; e += 448) {
        const int f = e & 31, tap = (e >> 5) & 15, p = e >> 9;
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < 7; ++wv) v += all[wv * SLAB + e];
        if (tap < K::TAPS && j0 + f < g.F && v != 0.f) atomicAdd(dw + (tap * 4 + p) * g.F + j0 + f, v);
    }
    if (dbias && tid < 128) {
        const int b = tid >> 5, f = tid & 31;
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < 7; ++wv) v += all[wv * SLAB + 4 * 16 * 32 + b * 64 + f];
        if (j0 + f < g.F) atomicAdd(dbias + b * g.F + j0 + f, v);
    }
}

template <typename T, bool PRELU>
int run_conv1_pool(bool backward, const void *x, const float *w, const float *bias, const void *io, const void *pre, void *argbits, float *dw,
                   float *dbias, float *dalpha, const C1Geom &g, hipStream_t stream)
{
    if (!backward) {
        int blocks = device_cu_count();                            // persistent, 159 - 239 registers: one 7-wave workgroup per CU
        if (blocks > g.n_lines) blocks = g.n_lines;
        dim3 grid((unsigned)blocks, (unsigned)((g.F + 31) / 32), 1);
        hipLaunchKernelGGL((k_conv1_pool_fwd<T, 3, 5, 3, PRELU>), grid, dim3(512), 0, stream, (const T *)x, w, bias, (T *)const_cast<void *>(io),
                           (T *)const_cast<void *>(pre), (uint2 *)argbits, g);
    } else {
        int blocks = device_cu_count();                            // persistent, > 200 registers: one 7-wave workgroup per CU
        if (blocks > g.n_lines) blocks = g.n_lines;
        if (debug_flags() & kDbgDeterministic) blocks = 1;         // one flush per gradient element: no order-dependent sums
        dim3 grid((unsigned)blocks, (unsigned)((g.F + 31) / 32), 1);
        hipLaunchKernelGGL((k_conv1_pool_bwd<T, 3, 5, 3, PRELU>), grid, dim3(448), 0, stream, (const T *)x, (const T *)io, (const T *)pre,
                           (const uint2 *)argbits, dw, dbias, dalpha, g);
    }
    return hipGetLastError() == hipSuccess ? 0 : QK_ERR_LAUNCH;
}

}  // namespace

// conv (3,5) 'same' on one quaternion channel + relu + max-pool (3,1) 'same' over H.  `argbits`: the arg-max side tensor
// (conv1_pool_argbits_bytes), written by the forward (may be NULL there: inference), read by the backward.
// alpha != NULL: the PReLU form (linear conv, slopes `alpha`, pre-activation of the arg-max in `pre`, slope gradient
// accumulated into `dalpha`).
size_t conv1_pool_argbits_bytes(int N, int H, int W, int F)
{
    const long long lines = (long long)N * ((H + 2) / 3) * ((W + C1_TW - 1) / C1_TW);
    return (size_t)(lines * ((F + 31) / 32) * 7 * 64 * 24);
}

int launch_conv1_pool(int dtype, bool backward, const void *x, const float *w, const float *bias, const void *io, void *argbits,
                      float *dw, float *dbias, int N, int H, int W, int F, int has_bias, hipStream_t stream,
                      const float *alpha, int alpha_len, const void *pre, float *dalpha, int x_planes)
{
    C1Geom g;
    g.N = N; g.H = H; g.W = W; g.F = F; g.has_bias = has_bias; g.x_planes = x_planes;
    g.Ho = (H + 2) / 3;
    g.n_chunks = (W + C1_TW - 1) / C1_TW;
    g.n_lines = N * g.Ho * g.n_chunks;
    g.alpha = alpha; g.alpha_len = alpha_len;
    if (alpha) {
        if (dtype == QK_BF16) return run_conv1_pool<bf16, true>(backward, x, w, bias, io, pre, argbits, dw, dbias, dalpha, g, stream);
        if (dtype == QK_F16) return run_conv1_pool<f16, true>(backward, x, w, bias, io, pre, argbits, dw, dbias, dalpha, g, stream);
        return QK_ERR_UNSUPPORTED;
    }
    if (dtype == QK_BF16) return run_conv1_pool<bf16, false>(backward, x, w, bias, io, nullptr, argbits, dw, dbias, nullptr, g, stream);
    if (dtype == QK_F16) return run_conv1_pool<f16, false>(backward, x, w, bias, io, nullptr, argbits, dw, dbias, nullptr, g, stream);
    return QK_ERR_UNSUPPORTED;
}

}  // namespace qk
