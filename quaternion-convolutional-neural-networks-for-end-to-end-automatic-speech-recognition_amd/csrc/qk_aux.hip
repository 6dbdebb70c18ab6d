// Small HBM-bound helpers: the fused Adam step and the tap-folding gather.
#include "qk_common.h"
#include "qk_postop.h"

namespace qk {
namespace {

// Keras-2 Adam (keras/optimizers.py Adam.get_updates), the optimiser of working_example.py:106.
// DECAY: the l2 kernel regularisers of the model (interspeech_model.py:63,68,173: loss += l2 * sum w^2) enter as
// g += decay[i] * p[i] with decay = 2 * l2 on regularised parameters, 0 elsewhere -- the term Keras' autodiff adds.
template <bool ZERO, bool DECAY>
__global__ void __launch_bounds__(256)
k_adam(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
       float *__restrict__ v, const float *__restrict__ decay, size_t n, float lr_t, float b1, float b2, float eps, float gscale,
       const int *__restrict__ step_dev, float lr)
{
    if (step_dev) {
        // the step number lives on the device (qk_adam_step_dev): *step_dev steps have been applied, this is step *step_dev + 1;
        // Keras' bias-corrected rate is formed here, in double as the host form does, so that no launch argument depends on the step
        __shared__ float lr_s;
        if (threadIdx.x == 0) {
            const double t = (double)(*step_dev + 1);
            lr_s = (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
        }
        __syncthreads();
        lr_t = lr_s;
    }
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        float gi = g[i] * gscale;
        if constexpr (DECAY) gi = fmaf(decay[i], p[i], gi);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
        if constexpr (ZERO) g[i] = 0.f;
    }
}
__global__ void k_bump(int *c) { *c += 1; }          // (behind k_adam on the same stream: every block has read the old value)

// Tap folding: xcol[m, a*cq2 + t*Cq + c] = x[pos(m,t), a*Cq + c]  (zero in the padding and beyond
// taps*Cq).  One thread per (row, component): the row is decoded once and the folded channels are
// walked with running (tap, channel) counters -- no division per element (the first version decoded
// row and tap for every group of 8 channels with runtime divisors and was VALU-bound: 620 us for the
// first TIMIT layer at B = 256, where the 537 MB it writes take ~135 us of HBM time).  HBM-bound: the
// output (M x 4*cq2) dominates, x itself stays in L2.
template <typename T> __device__ __forceinline__ void store8(T *dst, const T (&v)[8])
{
    if constexpr (sizeof(T) == 2) {
        *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(v);
    } else {
        *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4 *>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
k_fold_taps(const T *__restrict__ x, T *__restrict__ xcol, const GemmGeom g, int cq2)
{
    const int groups = cq2 / 8;
    const long long total = (long long)g.M * 4;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int a = (int)(idx & 3);
        int m = (int)(idx >> 2);
        T *dst = xcol + ((long long)m * 4 + a) * cq2;
        const int o2 = m % g.osp[2]; m /= g.osp[2];
        const int o1 = m % g.osp[1]; m /= g.osp[1];
        const int o0 = m % g.osp[0];
        const int n = m / g.osp[0];
        const int p0 = o0 * g.pa[0] + g.pc[0], p1 = o1 * g.pa[1] + g.pc[1], p2 = o2 * g.pa[2] + g.pc[2];
        const T *xa = x + (long long)n * g.in_sn + (long long)(a * g.Q) * g.in_sc;
        int t = 0, t0 = 0, t1 = 0, t2 = 0, c = 0;
        for (int k8 = 0; k8 < groups; ++k8) {
            __attribute__((aligned(16))) T v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                T val = from_f32<T>(0.f);
                if (t < g.taps) {
                    const int i0 = p0 + t0 * g.pb[0], i1 = p1 + t1 * g.pb[1], i2 = p2 + t2 * g.pb[2];
                    if (i0 >= 0 && i0 < g.isp[0] && i1 >= 0 && i1 < g.isp[1] && i2 >= 0 && i2 < g.isp[2])
                        val = xa[(long long)i0 * g.in_ss[0] + (long long)i1 * g.in_ss[1] + (long long)i2 * g.in_ss[2] +
                                 (long long)c * g.in_sc];
                }
                v[j] = val;
                if (++c == g.Q) {
                    c = 0; ++t;
                    if (++t2 == g.ks[2]) { t2 = 0; if (++t1 == g.ks[1]) { t1 = 0; ++t0; } }
                }
            }
            store8(dst + k8 * 8, v);
        }
    }
}

// data *= (mask > 0), elementwise (fallback of QK_BWD_MASK_DX for the kernels without an epilogue mask)
template <typename T>
__global__ void __launch_bounds__(256)
k_mask_gt0(T *__restrict__ data, const T *__restrict__ mask, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        if (!(to_f32(mask[i]) > 0.f)) data[i] = from_f32<T>(0.f);
}

// Max pooling, channels_last, non-overlapping windows.  One thread per (output position, group of V
// channels); HBM-bound: forward reads x once and writes y, backward reads x and dy and writes dx.
template <typename T, int V> struct VecOf;
template <> struct VecOf<bf16, 8> { typedef uint4 type; };
template <> struct VecOf<f16, 8> { typedef uint4 type; };
template <> struct VecOf<float, 4> { typedef float4 type; };

template <typename T, int V, bool BWD>
__global__ void __launch_bounds__(256)
k_maxpool(const T *__restrict__ x, const T *__restrict__ dy, T *__restrict__ out, const PoolGeom g)
{
    typedef typename VecOf<T, V>::type vec_t;
    const int cg = g.C / V;
    const long long total = (long long)g.batch * g.oh * g.ow * cg;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % cg) * V;
        long long r = idx / cg;
        const int ow = (int)(r % g.ow); r /= g.ow;
        const int oh = (int)(r % g.oh);
        const int n = (int)(r / g.oh);
        const int h0 = oh * g.wh, w0 = ow * g.ww;
        const int h1 = min(h0 + g.wh, g.ih), w1 = min(w0 + g.ww, g.iw);
        const T *xn = x + (long long)n * g.ih * g.iw * g.C + c;
        float best[V];
        int arg[V];
#pragma unroll
        for (int k = 0; k < V; ++k) { best[k] = -INFINITY; arg[k] = 0; }
        int pos = 0;
        for (int h = h0; h < h1; ++h)
            for (int w = w0; w < w1; ++w, ++pos) {
                __attribute__((aligned(16))) T v[V];
                *reinterpret_cast<vec_t *>(v) = *reinterpret_cast<const vec_t *>(xn + ((long long)h * g.iw + w) * g.C);
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const float f = to_f32(v[k]);
                    if (f > best[k] || f != f) { best[k] = f; arg[k] = pos; }    // first maximum wins, NaN propagates
                }
            }
        if constexpr (!BWD) {
            __attribute__((aligned(16))) T v[V];
#pragma unroll
            for (int k = 0; k < V; ++k) v[k] = from_f32<T>(best[k]);
            *reinterpret_cast<vec_t *>(out + (((long long)n * g.oh + oh) * g.ow + ow) * g.C + c) = *reinterpret_cast<const vec_t *>(v);
        } else {
            __attribute__((aligned(16))) T d[V];
            *reinterpret_cast<vec_t *>(d) = *reinterpret_cast<const vec_t *>(dy + (((long long)n * g.oh + oh) * g.ow + ow) * g.C + c);
            T *dn = out + (long long)n * g.ih * g.iw * g.C + c;
            pos = 0;
            for (int h = h0; h < h1; ++h)
                for (int w = w0; w < w1; ++w, ++pos) {
                    __attribute__((aligned(16))) T v[V];
#pragma unroll
                    for (int k = 0; k < V; ++k) v[k] = arg[k] == pos ? d[k] : from_f32<T>(0.f);
                    *reinterpret_cast<vec_t *>(dn + ((long long)h * g.iw + w) * g.C) = *reinterpret_cast<const vec_t *>(v);
                }
        }
    }
}

template <typename T, int V>
int run_maxpool(bool backward, const void *x, const void *dy, void *out, const PoolGeom &g, hipStream_t stream)
{
    const long long total = (long long)g.batch * g.oh * g.ow * (g.C / V);
    long long blocks = (total + 255) / 256;
    if (blocks > 262144) blocks = 262144;
    if (blocks < 1) blocks = 1;
    if (backward)
        hipLaunchKernelGGL((k_maxpool<T, V, true>), dim3((unsigned)blocks), dim3(256), 0, stream, (const T *)x, (const T *)dy, (T *)out, g);
    else
        hipLaunchKernelGGL((k_maxpool<T, V, false>), dim3((unsigned)blocks), dim3(256), 0, stream, (const T *)x, (const T *)dy, (T *)out, g);
    return hipGetLastError() == hipSuccess ? 0 : QK_ERR_LAUNCH;
}


// ---------------------------------------------------------------------------------------
// PReLU + Dropout on their own (qk_postop_fwd / qk_postop_bwd): any dtype, HBM-bound.  One thread handles VEC
// consecutive channels of one row; rows -> alpha index through (stride, extent) of the alpha axis.
// ---------------------------------------------------------------------------------------
struct PostDims { long long units; int upr; int key_div; int key_mod; };   // units = rows * upr, upr = units per row

template <typename T, bool BWD>
__global__ void __launch_bounds__(256)
k_postop(const T *__restrict__ pre, const T *__restrict__ dy, T *__restrict__ out, float *__restrict__ dalpha,
         const PostOp p_in, const PostDims d)
{
    const PostOp p = resolve_seed(p_in);
    constexpr int VEC = sizeof(T) == 2 ? 8 : 4;
    __shared__ float slab[256];
    const int tid = threadIdx.x, lane = tid & 63;
    if (BWD) { slab[tid] = 0.f; __syncthreads(); }
    for (long long base = (long long)blockIdx.x * 256; base < d.units; base += (long long)gridDim.x * 256) {
        const long long u = base + tid;
        const bool ok = u < d.units;
        int key = 0;
        float dal = 0.f;
        if (ok) {
            const long long row = u / d.upr;
            if (p.alpha_sel >= 0) key = (int)((row / d.key_div) % d.key_mod);
            const float alpha = p.alpha ? p.alpha[key] : 0.f;
            const long long e0 = u * VEC;
            if constexpr (sizeof(T) == 2) {
                const uint4 q = *reinterpret_cast<const uint4 *>(pre + e0);
                uint4 r;
                if constexpr (BWD) r = post_bwd8<T>(*reinterpret_cast<const uint4 *>(dy + e0), q, alpha, (unsigned)e0, p, dal);
                else r = post_fwd8<T>(q, alpha, (unsigned)e0, p);
                *reinterpret_cast<uint4 *>(out + e0) = r;
            } else {
                const float4 q = *reinterpret_cast<const float4 *>(pre + e0);
                float v[4] = {q.x, q.y, q.z, q.w}, g[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (BWD) { const float4 t = *reinterpret_cast<const float4 *>(dy + e0); g[0] = t.x; g[1] = t.y; g[2] = t.z; g[3] = t.w; }
                unsigned rb[2] = {0u, 0u};
                if (p.drop_thr) drop_bits8((unsigned)(e0 >> 3), p.drop_seed, rb[0], rb[1]);
                const unsigned bits = rb[(e0 >> 2) & 1];                   // this thread's 4 elements: half a unit
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    float ka = 1.f, kb = 1.f;
                    if (p.drop_thr) { ka = drop_factor(bits, 2 * k, p); kb = drop_factor(bits, 2 * k + 1, p); }
                    if constexpr (BWD) {
                        v[2 * k] = post_bwd1(g[2 * k] * ka, v[2 * k], alpha, dal);
                        v[2 * k + 1] = post_bwd1(g[2 * k + 1] * kb, v[2 * k + 1], alpha, dal);
                    } else {
                        v[2 * k] = post_fwd1(v[2 * k], alpha, ka);
                        v[2 * k + 1] = post_fwd1(v[2 * k + 1], alpha, kb);
                    }
                }
                *reinterpret_cast<float4 *>(out + e0) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        if constexpr (BWD) wave_add_by_key(dal, key, slab, lane);
    }
    if constexpr (BWD) {
        __syncthreads();
        if (tid < p.alpha_len && slab[tid] != 0.f) atomicAdd(dalpha + tid, slab[tid]);
    }
}

// ---------------------------------------------------------------------------------------
// Re-layout of a 16-bit activation between channels_first and channels_last: per sample a (A, B) matrix with B
// contiguous becomes (B, A) with A contiguous.  channels_first -> channels_last: A = channels, B = positions; the way
// back: A = positions, B = channels.  64 x 64 tiles through LDS: 16-byte global loads along B, 16-byte global stores
// along A (a tile row pitch of 33 words keeps the transposing 2-byte LDS reads on distinct banks).  HBM-bound.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_relayout16(const unsigned short *__restrict__ src, unsigned short *__restrict__ dst, int A, int B, int vec_ok)
{
    constexpr int TP = 66;                                  // pitch in elements (33 words)
    __shared__ unsigned short tile[64 * TP];
    const int tid = threadIdx.x;
    const long long n_off = (long long)blockIdx.z * A * B;
    const int b0 = blockIdx.x * 64, a0 = blockIdx.y * 64;
    const int v = tid & 7;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int r = (tid >> 3) + 32 * k;                  // row of the tile: index along A
        const int a = a0 + r, b = b0 + v * 8;
        unsigned w[4] = {0u, 0u, 0u, 0u};
        if (a < A) {
            const unsigned short *p = src + n_off + (long long)a * B + b;
            if (vec_ok && b + 8 <= B) {
                const uint4 q = *reinterpret_cast<const uint4 *>(p);
                w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (b + i < B) w[i >> 1] |= (unsigned)p[i] << (16 * (i & 1));
            }
        }
        unsigned *t4 = reinterpret_cast<unsigned *>(tile + r * TP + v * 8);
        t4[0] = w[0]; t4[1] = w[1]; t4[2] = w[2]; t4[3] = w[3];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int r = (tid >> 3) + 32 * k;                  // row of the transposed tile: index along B
        const int b = b0 + r, a = a0 + v * 8;
        if (b >= B) continue;
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            w[i] = (unsigned)tile[(v * 8 + 2 * i) * TP + r] | ((unsigned)tile[(v * 8 + 2 * i + 1) * TP + r] << 16);
        unsigned short *p = dst + n_off + (long long)b * A + a;
        if (vec_ok && a + 8 <= A) *reinterpret_cast<uint4 *>(p) = make_uint4(w[0], w[1], w[2], w[3]);
        else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (a + i < A) p[i] = (unsigned short)(w[i >> 1] >> (16 * (i & 1)));
        }
    }
}


// ---------------------------------------------------------------------------------------
// Row softmax behind the TIMIT model's TimeDistributed(Dense(62, activation='softmax')) (interspeech_model.py:171-175):
// one wave per row, lane j = class j (cols <= 64), fp32 logits in (the GEMM's fp32 output), T out.
//   fwd:  y = softmax(logits + bias)
//   bwd:  d logits = y * (dy - sum_j dy_j y_j)   (T, the operand of the two gradient GEMMs),
//         d bias[j] += sum over rows of d logits   (one atomic per class and workgroup)
// Replaces ~25 elementwise / reduction launches of framework glue per step (cast, bias add, softmax forward and
// backward in 16 bits, the bias-gradient reduction and the fills they need).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_max64(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum64(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <typename T>
__global__ void __launch_bounds__(256)
k_softmax_rows_fwd(const float *__restrict__ logits, const float *__restrict__ bias, T *__restrict__ y, long long rows, int cols)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    const bool live = lane < cols;
    const float b = (live && bias) ? bias[lane] : 0.f;
    for (long long r = wave; r < rows; r += nwaves) {
        const float v = live ? logits[r * cols + lane] + b : -INFINITY;
        const float m = wave_max64(v);
        const float e = live ? __expf(v - m) : 0.f;
        const float s = wave_sum64(e);
        if (live) y[r * cols + lane] = from_f32<T>(e / s);
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
k_softmax_rows_bwd(const T *__restrict__ y, const T *__restrict__ dy, T *__restrict__ dlogits, float *__restrict__ dbias,
                   long long rows, int cols)
{
    // (bias gradient: one atomic per class and workgroup.  The launcher keeps the grid at <= 512 workgroups: 3200 of them
    //  x 62 atomics on two cache lines cost 80 us; a last-workgroup-reduces scheme cost more still -- its agent-scope
    //  release per workgroup writes the L2 back 2048 times -- measured, round 4)
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long wave = (long long)blockIdx.x * 4 + w, nwaves = (long long)gridDim.x * 4;
    const bool live = lane < cols;
    float db = 0.f;
    for (long long r0 = wave * 4; r0 < rows; r0 += nwaves * 4) {          // four rows in flight per wave
        float yv[4], gv[4], dot[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool ok = live && r0 + k < rows;
            yv[k] = ok ? to_f32(y[(r0 + k) * cols + lane]) : 0.f;
            gv[k] = ok ? to_f32(dy[(r0 + k) * cols + lane]) : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) dot[k] = wave_sum64(yv[k] * gv[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const T q = from_f32<T>(yv[k] * (gv[k] - dot[k]));
            if (live && r0 + k < rows) dlogits[(r0 + k) * cols + lane] = q;
            db += to_f32(q);                          // the bias gradient is the column sum of what the GEMMs see
        }
    }
    if (!dbias) return;
    part[w][lane] = db;
    __syncthreads();
    if (w == 0 && live) atomicAdd(dbias + lane, part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]);
}

// sum_i a_i * w_i of a T tensor and an fp32 weight tensor, ADDED to *out (the bench's linear stand-in loss)
template <typename T>
__global__ void __launch_bounds__(256)
k_weighted_sum(const T *__restrict__ a, const float *__restrict__ w, float *__restrict__ out, long long n)
{
    __shared__ float part[4];
    float s = 0.f;
    const long long n4 = n / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 wv = reinterpret_cast<const float4 *>(w)[i];
        const T *ap = a + 4 * i;
        s = fmaf(to_f32(ap[0]), wv.x, s); s = fmaf(to_f32(ap[1]), wv.y, s); s = fmaf(to_f32(ap[2]), wv.z, s); s = fmaf(to_f32(ap[3]), wv.w, s);
    }
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n - 4 * n4)) s = fmaf(to_f32(a[4 * n4 + threadIdx.x]), w[4 * n4 + threadIdx.x], s);
    s = wave_sum64(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

}  // namespace

// (n, A, B) -> (n, B, A), 16-bit elements (see k_relayout16)
int launch_relayout16(const void *src, void *dst, int n, int A, int B, hipStream_t stream)
{
    if (n <= 0 || A <= 0 || B <= 0) return 0;
    const int vec_ok = (A % 8 == 0 && B % 8 == 0 && (reinterpret_cast<uintptr_t>(src) % 16) == 0 && (reinterpret_cast<uintptr_t>(dst) % 16) == 0) ? 1 : 0;
    if (n > 65535 || (A + 63) / 64 > 65535) return QK_ERR_UNSUPPORTED;
    dim3 grid((unsigned)((B + 63) / 64), (unsigned)((A + 63) / 64), (unsigned)n);
    hipLaunchKernelGGL(k_relayout16, grid, dim3(256), 0, stream, (const unsigned short *)src, (unsigned short *)dst, A, B, vec_ok);
    return hipGetLastError() == hipSuccess ? 0 : QK_ERR_LAUNCH;
}

int launch_mask_gt0(int dtype, void *data, const void *mask, size_t n, hipStream_t stream)
{
    if (n == 0) return 0;
    size_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    switch (dtype) {
    case QK_F32: hipLaunchKernelGGL(k_mask_gt0<float>, dim3((unsigned)blocks), dim3(256), 0, stream, (float *)data, (const float *)mask, n); break;
    case QK_BF16: hipLaunchKernelGGL(k_mask_gt0<bf16>, dim3((unsigned)blocks), dim3(256), 0, stream, (bf16 *)data, (const bf16 *)mask, n); break;
    case QK_F16: hipLaunchKernelGGL(k_mask_gt0<f16>, dim3((unsigned)blocks), dim3(256), 0, stream, (f16 *)data, (const f16 *)mask, n); break;
    default: return QK_ERR_INVALID_ARG;
    }
    return hipGetLastError() == hipSuccess ? 0 : QK_ERR_LAUNCH;
}

int launch_maxpool(int dtype, bool backward, const void *x, const void *dy, void *out, const PoolGeom &g, hipStream_t stream)
{
    switch (dtype) {
    case QK_F32: return run_maxpool<float, 4>(backward, x, dy, out, g, stream);
    case QK_BF16: return run_maxpool<bf16, 8>(backward, x, dy, out, g, stream);
    case QK_F16: return run_maxpool<f16, 8>(backward, x, dy, out, g, stream);
    default: return QK_ERR_INVALID_ARG;
    }
}

int launch_fold_taps(int dtype, const void *x, void *xcol, const GemmGeom &g, int cq2, hipStream_t stream)
{
    const long long total = (long long)g.M * 4;
    long long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    switch (dtype) {
    case QK_F32: hipLaunchKernelGGL(k_fold_taps<float>, dim3((unsigned)blocks), dim3(256), 0, stream, (const float *)x, (float *)xcol, g, cq2); break;
    case QK_BF16: hipLaunchKernelGGL(k_fold_taps<bf16>, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16 *)x, (bf16 *)xcol, g, cq2); break;
    case QK_F16: hipLaunchKernelGGL(k_fold_taps<f16>, dim3((unsigned)blocks), dim3(256), 0, stream, (const f16 *)x, (f16 *)xcol, g, cq2); break;
    default: return QK_ERR_INVALID_ARG;
    }
    return hipGetLastError() == hipSuccess ? 0 : QK_ERR_LAUNCH;
}

int launch_adam(float *p, float *g, float *m, float *v, const float *decay, size_t n, float lr, float b1,
                float b2, float eps, int step, float gscale, bool zero_grad, hipStream_t stream, int *step_dev)
{
    if (n == 0) return 0;
    const double t = (double)step;
    const float lr_t = step_dev ? 0.f : (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
#define QK_ADAM(Z, D) hipLaunchKernelGGL((k_adam<Z, D>), dim3((unsigned)blocks), dim3(256), 0, stream, p, g, m, v, decay, n, lr_t, b1, b2, eps, gscale, (const int *)step_dev, lr)
    if (zero_grad) { if (decay) QK_ADAM(true, true); else QK_ADAM(true, false); }
    else { if (decay) QK_ADAM(false, true); else QK_ADAM(false, false); }
#undef QK_ADAM
    if (step_dev) hipLaunchKernelGGL(k_bump, dim3(1), dim3(1), 0, stream, step_dev);
    return hipGetLastError() == hipSuccess ? 0 : QK_ERR_LAUNCH;
}


int launch_postop(int dtype, bool backward, const void *pre, const void *dy, void *out, float *dalpha, const PostOp &p,
                  long long rows, int channels, int key_div, int key_mod, hipStream_t stream)
{
    const int vec = dtype == QK_F32 ? 4 : 8;
    PostDims d;
    d.upr = channels / vec; d.units = rows * d.upr; d.key_div = key_div; d.key_mod = key_mod;
    long long blocks = (d.units + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
#define QK_PO(T, B) hipLaunchKernelGGL((k_postop<T, B>), dim3((unsigned)blocks), dim3(256), 0, stream, (const T *)pre, (const T *)dy, (T *)out, dalpha, p, d)
    if (dtype == QK_F32) { if (backward) QK_PO(float, true); else QK_PO(float, false); }
    else if (dtype == QK_BF16) { if (backward) QK_PO(bf16, true); else QK_PO(bf16, false); }
    else { if (backward) QK_PO(f16, true); else QK_PO(f16, false); }
#undef QK_PO
    return hipGetLastError() == hipSuccess ? 0 : QK_ERR_LAUNCH;
}

int launch_softmax_rows(int dtype, bool backward, const void *a, const void *b, void *out, float *dbias, long long rows, int cols,
                        hipStream_t stream)
{
    long long blocks = (rows + 15) / 16;              // four rows per wave at least
    // (backward: one atomic per class and workgroup; QK_DBG_DETERMINISTIC: ONE workgroup -- each bias column receives a single
    //  addition, its four wave partials summed in a fixed order)
    const long long cap = backward ? (((debug_flags() & kDbgDeterministic) && dbias) ? 1 : 512) : 2048;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
#define QK_SM(T) do { if (backward) hipLaunchKernelGGL((k_softmax_rows_bwd<T>), dim3((unsigned)blocks), dim3(256), 0, stream, (const T *)a, (const T *)b, (T *)out, dbias, rows, cols); \
                      else hipLaunchKernelGGL((k_softmax_rows_fwd<T>), dim3((unsigned)blocks), dim3(256), 0, stream, (const float *)a, (const float *)b, (T *)out, rows, cols); } while (0)
    if (dtype == QK_F32) QK_SM(float); else if (dtype == QK_BF16) QK_SM(bf16); else QK_SM(f16);
#undef QK_SM
    return hipGetLastError() == hipSuccess ? 0 : QK_ERR_LAUNCH;
}

int launch_weighted_sum(int dtype, const void *a, const float *w, float *out, long long n, hipStream_t stream)
{
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 512) blocks = 512;                   // one atomic per workgroup on ONE address: keep them few
    if (debug_flags() & kDbgDeterministic) blocks = 1;                  // ... and exactly one when the sum has to be bit-repeatable
    if (blocks < 1) blocks = 1;
    if (dtype == QK_F32) hipLaunchKernelGGL((k_weighted_sum<float>), dim3((unsigned)blocks), dim3(256), 0, stream, (const float *)a, w, out, n);
    else if (dtype == QK_BF16) hipLaunchKernelGGL((k_weighted_sum<bf16>), dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16 *)a, w, out, n);
    else hipLaunchKernelGGL((k_weighted_sum<f16>), dim3((unsigned)blocks), dim3(256), 0, stream, (const f16 *)a, w, out, n);
    return hipGetLastError() == hipSuccess ? 0 : QK_ERR_LAUNCH;
}

}  // namespace qk
