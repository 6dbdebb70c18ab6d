// Internal definitions shared by the gfx950 kernels and the C-ABI front end (include/qk.h).
// CDNA4 only: 64-wide wavefronts, MFMA, LDS.  No portability layer.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/qk.h"

namespace qk {

// ---------------------------------------------------------------------------------------
// Hamilton sign tables.  Bit (a*4 + b) set  <=>  the product of GATHERED component a with
// compact part (a ^ b) enters PRODUCED component b with a minus sign.
//   conv.py:327-331   rows a (input), cols b (output):  [[+,+,+,+],[-,+,+,-],[-,-,+,+],[-,+,-,+]]
//   dense.py:139-143  is the transpose.
// ---------------------------------------------------------------------------------------
constexpr unsigned kSignConv = 0x5390u;   // negatives at (1,0) (2,0) (3,0) (2,1) (3,2) (1,3)
constexpr unsigned kSignConj = 0x284Eu;   // transpose: (0,1) (0,2) (0,3) (1,2) (2,3) (3,1)

// ---------------------------------------------------------------------------------------
// Implicit-GEMM geometry shared by forward and backward-data.
//
//   Out[m, b*J + j] = act( bias[b*J+j] + sum_{t, a, q} sgn(a,b) * In[pos(m,t), a*Q + q]
//                                                      * Wk[((t*Q + q)*4 + (a^b))*J + j] )
//
//   forward : In = x,  Q = Cq, J = F,  Wk = the compact kernel as stored (conv.py:165)
//   bwd-data: In = dy, Q = F,  J = Cq, Wk = the same compact kernel read with q/j swapped
//             (w_swapped), sign table transposed, and the position map inverted (pd = stride).
//
// pos(m, t): m -> (n, o0, o1, o2); per axis num = o*pa + t*pb + pc; the tap contributes iff
// num >= 0, num % pd == 0 and num/pd < isp (zero padding otherwise).
// ---------------------------------------------------------------------------------------
// PReLU (+ Dropout) attached to a layer output, see qk_postop.h and qk_postop_t (include/qk.h)
struct PostOp {
    int kind;               // 0: none, 1: y = drop(prelu(pre)) (pre and y are both kept), 2: y = drop(relu(pre)): alpha is
                            // NULL, only y is written and the backward needs only y (y > 0 <=> pre > 0 and kept)
    int alpha_sel;          // -1: scalar alpha[0]; 0 / 1 / 2: alpha indexed by the row's o0 / o1 / o2 (the kernel's axes)
    int alpha_len;          // entries of alpha (<= 256 on the fused paths)
    const float *alpha;     // device, float32
    float drop_scale;       // 1 / (1 - rate); 1 without dropout
    unsigned drop_thr;      // keep iff the element's 8 random bits >= drop_thr (= round(rate * 256)); 0 = no dropout
    unsigned drop_seed;
    const unsigned *seed_dev; // optional device counter mixed into drop_seed by the kernels (qk_postop_t.drop_seed_dev: graph replay)
};

struct GemmGeom {
    int M;              // batch * prod(osp)
    int batch;
    int osp[3];         // extents m decomposes into (the produced tensor's spatial extents)
    int isp[3];         // spatial extents of the gathered tensor
    int Q;              // gathered channels per component
    int J;              // produced channels per component
    int Qp, Jp;         // 16-bit MFMA path: Q / J rounded up to the kernels' 32-channel granule (the re-laid-out kernel is zero-padded to them)
    int ks[3];
    int taps;
    int pa[3], pb[3], pc[3], pd[3];
    long long in_sn, in_ss[3], in_sc;     // element strides of the gathered tensor
    long long out_sn, out_ss, out_sc;     // produced tensor: n*out_sn + s*out_ss + ch*out_sc
    unsigned sign_tbl;
    int relu;           // epilogue activation
    int has_bias;
    int has_mask;       // gathered value is zeroed where mask[same index] <= 0 (relu backward)
    int ablate;         // profiling only (env QK_ABLATE): 4 = skip the MFMA loop, 8 = skip the epilogue, 64 = band kernels with ONE workgroup per CU (probe builds: 16 / 32, qk_hgemm_bf16mfma.hip)
    unsigned b_rep;     // band kernels: sum over outer tap t0 of 2^(t0 * ks[1]) (0: more than 32 outer taps -- loop form)
    unsigned long long *dbg_ts;          // profiling only (qk_set_debug_buffer): per-workgroup phase time stamps of the band kernel
    int w_swapped;      // Wk is the compact kernel itself, read with q/j swapped (backward-data): no transposed copy
    int w_prepped;      // 16-bit path: the workspace already holds the re-laid-out kernel (qk_conv_desc_t.ws_has_kernel)
    // band variant of the 16-bit kernel (qk_hgemm_bf16mfma.hip): rows of M run over PADDED lines of the
    // innermost axis (b_wp = out extent + k - 1 positions per line, b_nlines lines), band row j of a tile
    // holds input position (padded position + b_cshift); b_rev: taps walk the band backwards (bwd-data)
    // division by the row-decode divisors through precomputed multipliers (set_fastdiv below): an integer
    // division is ~40 VALU instructions, a tile's prologue decodes 3 - 5 rows with 3 of them each
    unsigned dv_mul[3], dv_shr[3];       // k_hgemm16: by osp[2], osp[1], osp[0];  band: by b_wp, osp[1], osp[0]
    const void *ep_mask;                 // optional epilogue mask: out *= (ep_mask > 0), same layout as out (16-bit kernels)
    // post-op (16-bit kernels).  forward: out = post(pre), pre = bias + conv is ALSO written to pre_out;
    // backward-data: ep_mask holds the pre-activation of the tensor whose gradient is produced, out = d pre,
    // d alpha is accumulated into dalpha
    PostOp post;
    int post_fwd;                        // the post-op is applied to THIS kernel's output (forward); else (ep_mask set) its derivative
    void *pre_out;                       // forward, kind 1: where the pre-activation goes (kind 2 writes y only)
    float *dalpha;
    int b_wp, b_nlines, b_cshift, b_rev;
    // k_hgemm16 row order: 0 = rows run over (n, o0, o1, o2); batch = rows run over (o0, n, o1, o2) (dv_*[2] then divides
    // by batch).  Chosen when the gathered tensor has ONE position along axis 0 under a multi-tap kernel axis (the
    // backward-data of the (F, 1) 'valid' head convolution): the single valid tap of a row is then a function of o0
    // alone and every 128-row tile uses exactly one tap instead of the two a line boundary inside the tile brings.
    int o0_major;
    unsigned b_in_bytes, b_w_bytes;      // extents of the input tensor and of the re-laid-out kernel (buffer resources)
    int w_ch_major;     // qk_conv_desc_t.kernel_order == QK_KERNEL_CHANNEL_MAJOR: the compact kernel lies as (cq, taps, 4 fq); honoured by the 16-bit re-layout only
};

// Backward-weight geometry:  dW[t, c, p, f] = sum_{a^b=p} sgn(a,b) sum_m x_a[pos(m,t), c] * dy_b[m, f]
// Geometry of the band variants of the forward / backward-data kernels (qk_hgemm_bf16mfma.hip,
// qk_hgemm_f32mfma.inc), or false when the shape is outside them: the innermost used axis must have
// unit stride and dilation and 3 or 5 taps, no relu mask to apply, unit-stride position map, and the
// padding positions must stay a small part of the work.  Axes are rotated so that this axis is index 2
// (unit axes move to the front: neither the row order nor the tap order changes).  `esize` = bytes per
// activation element (extents for the buffer resources of the 16-bit kernel).
inline bool band_geom(const GemmGeom &g, int esize, GemmGeom *o)
{
    if (g.has_mask) return false;
    if (g.pd[0] != 1 || g.pd[1] != 1 || g.pd[2] != 1) return false;
    int ax = 2;
    while (ax > 0 && g.osp[ax] == 1 && g.isp[ax] == 1 && g.ks[ax] == 1) --ax;
    if (g.ks[ax] != 3 && g.ks[ax] != 5) return false;
    if (g.pa[ax] != 1 || (g.pb[ax] != 1 && g.pb[ax] != -1)) return false;
    *o = g;
    const int sh = 2 - ax;                                  // rotate axes right by sh
    for (int i = 0; i < 3; ++i) {
        const int src = i - sh;
        o->osp[i] = src >= 0 ? g.osp[src] : 1; o->isp[i] = src >= 0 ? g.isp[src] : 1; o->ks[i] = src >= 0 ? g.ks[src] : 1;
        o->pa[i] = src >= 0 ? g.pa[src] : 1; o->pb[i] = src >= 0 ? g.pb[src] : 1; o->pc[i] = src >= 0 ? g.pc[src] : 0;
        o->in_ss[i] = src >= 0 ? g.in_ss[src] : 0;
    }
    const int k = o->ks[2];
    if (o->ks[0] * o->ks[1] > 32) return false;             // outer-tap bit mask
    for (int i = 0; i < 2; ++i)                              // ... built from tap RANGES: dilation +-1 on the outer axes (else the general kernel)
        if (o->pb[i] != 1 && o->pb[i] != -1) return false;
    o->b_rep = 0;
    for (int t0 = 0; t0 < o->ks[0]; ++t0) o->b_rep |= 1u << (t0 * o->ks[1]);
    o->b_wp = o->osp[2] + k - 1;
    if ((k - 1) * 12 > o->b_wp) return false;                // > 8 % of the rows would be padding
    const long long lines = (long long)g.batch * o->osp[0] * o->osp[1];
    if (lines * o->b_wp >= (1ll << 31) - 512) return false;
    o->b_nlines = (int)lines;
    const long long Qw = g.Qp ? g.Qp : g.Q, Jw = g.Jp ? g.Jp : g.J;             // (padded extents of the re-laid-out kernel, 16-bit path)
    const long long in_bytes = (long long)g.batch * g.in_sn * esize, w_bytes = (long long)g.taps * Qw * 4 * Jw * 2 + 256;
    if (in_bytes >= 0xF0000000ll || w_bytes >= 0xF0000000ll) return false;   // 32-bit buffer offsets
    o->b_in_bytes = (unsigned)in_bytes;
    o->b_w_bytes = (unsigned)w_bytes;
    o->b_rev = o->pb[2] < 0;
    o->b_cshift = o->b_rev ? o->pc[2] - (k - 1) : o->pc[2];
    if (o->post.alpha_sel >= 0) o->post.alpha_sel += sh;     // the alpha axis moves with the rotation
    return true;
}

inline int pad32(int v) { return (v + 31) / 32 * 32; }

// n / d for 0 <= n < 2^31 with (mul, shr) from fastdiv_of(d): umulhi(n, mul) >> shr  (d == 1: mul == 0 marks identity)
inline void fastdiv_of(unsigned d, unsigned *mul, unsigned *shr)
{
    if (d <= 1) { *mul = 0; *shr = 0; return; }
    unsigned lg = 0;
    while ((1ull << lg) < d) ++lg;                         // ceil(log2 d)
    const unsigned p = 31 + lg;
    *mul = (unsigned)(((1ull << p) + d - 1) / d);
    *shr = p - 32;
}
__device__ __forceinline__ int fastdiv(int n, unsigned mul, unsigned shr)
{
    // branch-free: `mul` is wave-uniform, and hipcc turns `mul ? ... : n` into a scalar BRANCH around the multiply -- three
    // per decoded row in the prologues, which are instruction-issue bound (tools/probe/phase_stamps.py).  One bit-select instead.
    const unsigned q = __umulhi((unsigned)n, mul) >> shr;
    const unsigned id = (unsigned)-(int)(mul == 0u);            // all ones for the identity divisor
    return (int)((q & ~id) | ((unsigned)n & id));
}

// Bit (t0 * ks1 + t1) set iff outer tap (t0, t1) of a row whose tap-0 input coordinates are (q0, q1) falls inside the tensor.
// With dilation +-1 on both outer axes the valid taps of an axis are a RANGE, so the mask is a product of two bit ranges --
// no loop, no branch (the loops with their run-time trip counts were ~40 taken branches per wave in the band kernel's prologue,
// which is instruction-issue bound: tools/probe/phase_stamps.py).  `rep` = sum over t0 of 2^(t0 * ks1) (host: GemmGeom::b_rep).
__device__ __forceinline__ unsigned outer_tap_mask(int q0, int q1, const GemmGeom &g)
{
    // input coordinate q + pb t with pb = +1 (forward) or -1 (backward-data) -- band_geom admits nothing else on the outer axes.
    // pb = -1 is pb = +1 on the mirrored coordinate isp - 1 - q: ONE multiply-add by wave-uniform values instead of selects on a
    // uniform condition (which hipcc compiles to scalar branches); taps [lo, hi) are then inside [0, extent).
    const int m0 = q0 * g.pb[0] + (g.pb[0] > 0 ? 0 : g.isp[0] - 1), m1q = q1 * g.pb[1] + (g.pb[1] > 0 ? 0 : g.isp[1] - 1);
    const int lo0 = max(0, -m0), hi0 = min(g.ks[0], g.isp[0] - m0);
    const int lo1 = max(0, -m1q), hi1 = min(g.ks[1], g.isp[1] - m1q);
    const unsigned m1 = ((hi1 >= 32 ? 0u : (1u << hi1)) - 1u) & ~((1u << lo1) - 1u);
    const int b0 = lo0 * g.ks[1], e0 = hi0 * g.ks[1];
    const unsigned r0 = ((e0 >= 32 ? 0u : (1u << e0)) - 1u) & ~((1u << b0) - 1u);
    return (lo0 >= hi0 || lo1 >= hi1) ? 0u : (m1 * g.b_rep) & r0;
}


struct WgradGeom {
    int M;
    int batch;
    int osp[3];         // dy spatial extents (m decomposes into these)
    int isp[3];         // x spatial extents
    int Cq, F;
    int ks[3];
    int taps;
    int pa[3], pb[3], pc[3];              // forward position map (pd == 1)
    long long x_sn, x_ss[3], x_sc;
    long long dy_sn, dy_ss, dy_sc;
    unsigned sign_tbl;
    int has_mask;
    int want_dbias;
    int m_per_split;    // rows of M each split reduces (multiple of the kernel's K step)
    int n_splits;
    int ablate;         // profiling only (env QK_ABLATE): 1 = skip fold + atomics, 2 = skip HBM atomics
    int deterministic;  // QK_DBG_DETERMINISTIC: n_splits == 1 and ONE owner block per bias column / masked-dY row
    unsigned x_bytes, dy_bytes;   // extents of x and of dy / y / dym (buffer-resource bounds of the 16-bit kernel)
    // band variant (qk_wgrad_band_bf16mfma.hip): positions run over padded lines of the innermost axis (b_wp per
    // line, b_nlines lines); band row j of a tile holds input column (padded position + b_cshift)
    int b_wp, b_nlines, b_cshift;
    int w_ch_major;     // dw lies as (cq, taps, 4 fq) (qk_conv_desc_t.kernel_order): k_wgrad16 only
    void *dym;          // optional output: dy with the relu mask applied (same layout/dtype as dy), or NULL;
                        // written by the blocks of tap 0 / channel chunk 0, which see every (row, filter) once
    unsigned fd_mul[4], fd_sh[4];   // k_wgrad (fp32 MFMA): exact division of a row index by osp[2], osp[1], osp[0], S (make_fastdiv; set by the launcher)
};

// floor(n / d) for 0 <= n < 2^31 as one 32 x 32 -> 64 multiplication: mul = ceil(2^(31 + sh) / d), sh = ceil(log2 d)
inline void make_fastdiv(unsigned d, unsigned *mul, unsigned *sh)
{
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    *mul = (unsigned)(((1ull << (31 + s)) + d - 1) / d);
    *sh = s;
}
__host__ __device__ __forceinline__ int fast_div(int n, unsigned mul, unsigned sh)
{
    return (int)(((unsigned long long)(unsigned)n * mul) >> (31 + sh));
}

// ---------------------------------------------------------------------------------------
// Element types in HBM.  Accumulation is always fp32.
// ---------------------------------------------------------------------------------------
struct bf16 { uint16_t x; };
typedef _Float16 f16;

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return __uint_as_float(((uint32_t)v.x) << 16); }
__device__ __forceinline__ float to_f32(f16 v) { return (float)v; }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v)
{
    // round-to-nearest-even, NaN kept quiet
    uint32_t u = __float_as_uint(v);
    bf16 r;
    if ((u & 0x7fffffffu) > 0x7f800000u) { r.x = (uint16_t)((u >> 16) | 0x40u); return r; }
    u += 0x7fffu + ((u >> 16) & 1u);
    r.x = (uint16_t)(u >> 16);
    return r;
}
template <> __device__ __forceinline__ f16 from_f32<f16>(float v) { return (f16)v; }

// 4 consecutive elements -> 4 floats (16 B for f32, 8 B for the 16-bit types)
__device__ __forceinline__ void load4(const float *p, float (&o)[4])
{
    const float4 v = *reinterpret_cast<const float4 *>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ __forceinline__ void load4(const bf16 *p, float (&o)[4])
{
    const uint2 v = *reinterpret_cast<const uint2 *>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
}
__device__ __forceinline__ void load4(const f16 *p, float (&o)[4])
{
    typedef f16 h4 __attribute__((ext_vector_type(4)));
    const h4 v = *reinterpret_cast<const h4 *>(p);
    o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3];
}

// One element of the k_hconv16_small kernel layout (qk_hconv16_small.hip): [K step kk = outer tap x KSO + i][part p][16-filter block fb]
// [lane 64][8] -- exactly the A-operand fragments of v_mfma_f32_16x16x32 (lane: filter row lane & 15 of the block, K values
// 8 (lane >> 4) .. + 7 of the step), so a wave reads a fragment with ONE linear ds_read_b128.  A K step holds 32 gathered channels of
// ONE inner tap (Q = 32) or 16 channels of TWO inner taps (Q = 16; the odd tap out of 3 / 5 pairs with a zero phantom).
// Same sign fold / transposition conventions as k_prep_w16.
template <typename T>
__device__ __forceinline__ void prep_small16_write(const float *w, T *wq, long long idx, int Cq, int F, int transposed, int neg_ijk,
                                                   int kin, int q32, int fb_n)
{
    const int kso = q32 ? kin : (kin + 1) / 2;
    long long r = idx;
    const int e = (int)(r % 8); r /= 8;
    const int lane = (int)(r % 64); r /= 64;
    const int fb = (int)(r % fb_n); r /= fb_n;
    const int p = (int)(r % 4); r /= 4;
    const int kk = (int)r;
    const int k = (lane >> 4) * 8 + e;
    const int ot = kk / kso, ki = kk - ot * kso;
    const int tap_in = q32 ? ki : 2 * ki + (k >> 4);
    const int q = q32 ? k : (k & 15);
    const int j = fb * 16 + (lane & 15);
    float v = 0.f;
    if (tap_in < kin) {
        const int tap = ot * kin + tap_in;
        const int c = transposed ? j : q, f = transposed ? q : j;
        v = w[((long long)(tap * Cq + c) * 4 + p) * F + f];
        if (neg_ijk && p) v = -v;
    }
    wq[idx] = from_f32<T>(v);
}

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// row of accumulator register r of a 32x32 MFMA tile held by `lane` (column = lane & 31)
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---------------------------------------------------------------------------------------
// host-side launch entry points implemented in the .hip files
// ---------------------------------------------------------------------------------------
// fp32-MFMA Hamilton implicit GEMM (forward and backward-data)
int launch_hgemm(int dtype, const void *in, const void *mask, const float *wk, const float *bias,
                 void *out, const GemmGeom &g, bool vec_ok, hipStream_t stream);
// fp32-MFMA Hamilton backward-weight (+ fused bias gradient); dw/dbias must be zeroed before
int launch_wgrad(int dtype, const void *x, const void *dy, const void *ymask, float *dw,
                 float *dbias, WgradGeom g, bool vec_ok, hipStream_t stream);
int launch_fold_taps(int dtype, const void *x, void *xcol, const GemmGeom &g, int cq2, hipStream_t stream);
int launch_mask_gt0(int dtype, void *data, const void *mask, size_t n, hipStream_t stream);   // data *= (mask > 0)
// batched 16-bit re-layout of compact kernels (qk_conv_prep_kernels): up to 32 jobs per launch, passed by value
struct PrepJob { const float *w; void *wq; int taps, cq, fq, transposed, neg_ijk;
                 int small, kin, n_ot, ch_major; };      // small != 0: the fragment layout of k_hconv16_small (qk_hconv16_small.hip) -- kin inner taps, n_ot outer taps
struct PrepJobs { PrepJob j[32]; };
int launch_prep_w16_batch(int dtype, const PrepJobs &jobs, int n, hipStream_t stream);
// Small-channel 16-bit layers (Q, J in {16, 32}, not both 32: start_filter = 16 models) -- qk_hconv16_small.hip.
// small16_shape: pure geometry (what decides the LAYOUT of the cached 16-bit kernel: qk_conv_prep_kernels and the call agree by
// construction); *bg receives the band geometry.
struct Small16 { int q32, fb, kin, kso, n_ot, nkk; unsigned w_bytes; };
inline bool small16_shape(const GemmGeom &g_in, GemmGeom *bg, Small16 *s)
{
    // (Q, J) = (16, 16) and (32, 16).  (16, 32) -- two 16-filter blocks per wave, one workgroup per CU -- measured SLOWER than the band
    // kernel's PAD form (16 -> 32 forward at B = 256: 205 vs 171 - 199 us) and stays there; the kernel's FB = 2 instantiations remain
    // for the day its staging runs deeper
    if (!((g_in.Q == 16 || g_in.Q == 32) && g_in.J == 16)) return false;
    GemmGeom g = g_in;
    g.Qp = pad32(g.Q); g.Jp = pad32(g.J);
    if (!band_geom(g, 2, bg)) return false;           // 3 / 5 unit-stride inner taps, no relu mask on the gathered tensor, <= 32 outer taps ...
    s->q32 = g.Q == 32; s->fb = g.J / 16; s->kin = bg->ks[2]; s->n_ot = bg->ks[0] * bg->ks[1];
    s->kso = s->q32 ? s->kin : (s->kin + 1) / 2;
    s->nkk = s->n_ot * s->kso;
    s->w_bytes = (unsigned)(s->nkk * 4 * s->fb * 1024);
    return s->n_ot <= 3;                               // the whole kernel is resident in LDS (static: three outer taps) beside two band buffers
}
// bytes of the workspace region that holds the fragment layout (it sits BEHIND the band layout + zero line of the same kernel)
inline size_t small16_region_bytes(const Small16 &s) { return ((size_t)s.w_bytes + 255) / 256 * 256 + 256; }
int launch_prep_small16(int dtype, const float *w, void *wq, int Cq, int F, int transposed, int neg_ijk, const Small16 &s, hipStream_t stream);
int launch_hconv16_small(int dtype, const void *in, const void *wq, const float *bias, void *out, const GemmGeom &bg, const Small16 &s, hipStream_t stream);

size_t ctc_workspace_bytes(int B, int T, int Lmax);
int launch_ctc(int dtype, int B, int T, int C, const void *pred, const int *labels, int Lmax, const int *in_len, const int *lab_len,
               float *cost, void *dpred, float *ws, hipStream_t stream);
int launch_relayout16(const void *src, void *dst, int n, int A, int B, hipStream_t stream);     // (n, A, B) -> (n, B, A), 16-bit
struct PoolGeom { int batch, ih, iw, C, wh, ww, oh, ow; };
int launch_maxpool(int dtype, bool backward, const void *x, const void *dy, void *out, const PoolGeom &g, hipStream_t stream);
size_t conv1_pool_argbits_bytes(int N, int H, int W, int F);
int launch_conv1_pool(int dtype, bool backward, const void *x, const float *w, const float *bias, const void *io, void *argbits,
                      float *dw, float *dbias, int N, int H, int W, int F, int has_bias, hipStream_t stream,
                      const float *alpha = nullptr, int alpha_len = 0, const void *pre = nullptr, float *dalpha = nullptr,
                      int x_planes = 0);
int launch_postop(int dtype, bool backward, const void *pre, const void *dy, void *out, float *dalpha, const PostOp &p,
                  long long rows, int channels, int key_div, int key_mod, hipStream_t stream);
int launch_softmax_rows(int dtype, bool backward, const void *a, const void *b, void *out, float *dbias, long long rows, int cols,
                        hipStream_t stream);
int launch_weighted_sum(int dtype, const void *a, const float *w, float *out, long long n, hipStream_t stream);
// Dense(units, softmax) on a 16-bit (rows, K) matrix with fp32 master weights (qk_out_layer.hip)
bool dense_softmax_supported(int dtype, long long rows, int K, int U);
int launch_dense_softmax_fwd(int dtype, long long rows, int K, int U, const void *x, const float *w, const float *bias, void *y, hipStream_t stream);
size_t dense_softmax_bwd_workspace_bytes(int dtype, long long rows, int K, int U);
int launch_dense_softmax_bwd(int dtype, long long rows, int K, int U, const void *x, const float *w, const void *y, const void *dy, void *dx,
                             float *dw, float *dbias, const float *dy_scale_dev, float dy_scale, float *ws, hipStream_t stream);
int launch_adam(float *p, float *g, float *m, float *v, const float *decay, size_t n, float lr, float b1,
                float b2, float eps, int step, float gscale, bool zero_grad, hipStream_t stream, int *step_dev = nullptr);

void set_error(const char *fmt, ...);
void note_path(int qk_path);          // thread-local record behind qk_last_path()

// Diagnostic switches: one process-wide bit mask, initialised ONCE from the environment (QK_NO_MFMA16,
// QK_NO_BAND16, QK_NO_BAND32, QK_WGRAD16_ONE_TAP, QK_ABLATE=<bits>, QK_FORCE_CFG="policy,bq") and changeable at
// run time through qk_set_debug_flags() (include/qk.h).  Launch paths read one relaxed atomic -- no getenv, no
// other mutable global.  Not part of the compute contract: every combination computes the same values.
enum : unsigned {
    kDbgNoMfma16 = QK_DBG_NO_MFMA16, kDbgNoBand16 = QK_DBG_NO_BAND16, kDbgNoBand32 = QK_DBG_NO_BAND32,
    kDbgWgradOneTap = QK_DBG_WGRAD16_ONE_TAP, kDbgBand8Waves = QK_DBG_BAND16_8WAVES, kDbgNoWgradBand = QK_DBG_NO_WGRAD_BAND,
    kDbgNoPoint16 = QK_DBG_NO_POINT16, kDbgCtcTwoSweeps = QK_DBG_CTC_TWO_SWEEPS, kDbgDeterministic = QK_DBG_DETERMINISTIC, kDbgWgradBandV1 = QK_DBG_WGRAD_BAND_V1, kDbgNoSmall16 = QK_DBG_NO_SMALL16, kDbgAblateShift = 8, kDbgAblateMask = 0xffu << 8
};
unsigned debug_flags();
unsigned long long *debug_buffer(size_t *bytes);      // qk_set_debug_buffer (qk_api.hip)
inline int debug_ablate() { return (int)((debug_flags() & kDbgAblateMask) >> kDbgAblateShift); }
// QK_FORCE_CFG (fp32 tiling sweep, tools/gpu_cfgsweep.sh): parsed once; false when unset
bool debug_force_cfg(int *policy, int *bq);
// compute units x resident workgroups of a kernel, queried per call from the current device (no caching)
int device_cu_count();

}  // namespace qk
