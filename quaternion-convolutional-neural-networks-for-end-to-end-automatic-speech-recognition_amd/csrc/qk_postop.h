// Post-op of a quaternion layer inside the TIMIT model: keras PReLU followed by Dropout
// (models/interspeech_model.py:99-101,117-121 of the reference: `O = PReLU(shared_axes=[1,0])(O); O = Dropout(d.dropout)(O)`)
//
//     y = drop(prelu(pre)),   prelu(v) = v > 0 ? v : alpha * v,   drop(v) = keep ? v / (1 - rate) : 0
//
// and, for the `aact == 'none'` setting of the same model (relu layers, `O = Dropout(d.dropout)(O)` behind every body
// convolution, :117-121,131-137), kind 2:  y = drop(relu(pre))  -- alpha == 0 by definition, ONE output tensor, and
// the backward reads nothing but y:  d pre = dy / (1 - rate) where y > 0  (y > 0  <=>  pre > 0 and kept).
//
// fused into the epilogue of the kernel that produces `pre` (forward: both `pre` and `y` are written) and, in the
// backward, into the epilogue of the NEXT layer's backward-data kernel, whose output IS the gradient w.r.t. y:
//
//     g = drop'(dy_out),   d pre = pre > 0 ? g : pre < 0 ? alpha * g : 0,   d alpha += sum over (pre < 0) of g * pre
//
// alpha is either one scalar or one slope per position along ONE spatial axis of the layer output (what Keras'
// `shared_axes=[1, 0]` yields for a channels_first (C, F, T) tensor: (1, F, 1), see layers.PReLU).  The dropout mask
// is not stored: it is a counter-based hash of (seed, flat element index of y in its channels-last buffer), evaluated
// again in the backward.  One hash (+ one extra mixing round) serves the 8 elements of a 16-byte unit, 8 bits each:
// keep iff bits >= round(rate * 256).
#pragma once
#include "qk_common.h"

namespace qk {

// 64 random bits for the 8 elements of unit `unit_index` (= flat index / 8): 8 bits per element, keep iff >= drop_thr
__device__ __forceinline__ void drop_bits8(unsigned unit_index, unsigned seed, unsigned &lo, unsigned &hi)
{
    unsigned h = (unit_index ^ seed) * 0x9E3779B1u;
    h ^= h >> 15; h *= 0x85EBCA77u;
    h ^= h >> 13; h *= 0xC2B2AE3Du;
    lo = h ^ (h >> 16);
    hi = (lo ^ 0x68E31DA4u) * 0xB5297A4Du;
    hi ^= hi >> 15;
}
// The seed the hash runs on: drop_seed, plus -- when the caller gave a device counter (qk_postop_t.drop_seed_dev) -- a multiple of
// the counter's CURRENT value: a captured graph replays the same launch arguments every step and still draws new masks, the
// forward and the backward of one step agree because the counter moves only between steps (qk_adam_step_dev).  Resolved ONCE at
// kernel entry into a by-value copy of the post-op (one scalar load; a load at the point of use sat in the epilogues' inner
// branches as a vector load with a vmcnt(0) behind it).
__device__ __forceinline__ PostOp resolve_seed(const PostOp &p)
{
    PostOp r = p;
    if (r.seed_dev) r.drop_seed += __builtin_amdgcn_readfirstlane(*r.seed_dev) * 0x9E3779B1u;
    r.seed_dev = nullptr;
    return r;
}
// scale factor of element e (0..3) of the half-unit whose bits are `bits`: 0 when dropped
__device__ __forceinline__ float drop_factor(unsigned bits, int e, const PostOp &p)
{
    return ((bits >> (8 * e)) & 0xffu) >= p.drop_thr ? p.drop_scale : 0.f;
}

template <typename T> __device__ __forceinline__ void unpack2(unsigned u, float &a, float &b);
template <> __device__ __forceinline__ void unpack2<bf16>(unsigned u, float &a, float &b)
{
    a = __uint_as_float(u << 16); b = __uint_as_float(u & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack2<f16>(unsigned u, float &a, float &b)
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 v = __builtin_bit_cast(h2, u);
    a = (float)v[0]; b = (float)v[1];
}
typedef float pf32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned repack2(bf16, float a, float b)
{
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const pf32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));
}
__device__ __forceinline__ unsigned repack2(f16, float a, float b)
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const pf32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2));
}

__device__ __forceinline__ float post_fwd1(float v, float alpha, float keep)
{
    return (fmaxf(v, 0.f) + alpha * fminf(v, 0.f)) * keep;
}
// d pre and the slope-gradient term of one element; g = dy * keep
__device__ __forceinline__ float post_bwd1(float g, float pre, float alpha, float &dal)
{
    dal = fmaf(g, fminf(pre, 0.f), dal);
    return g * (pre > 0.f ? 1.f : (pre < 0.f ? alpha : 0.f));
}

// 8 consecutive 16-bit elements starting at flat index `idx` (a multiple of 8)
template <typename T>
__device__ __forceinline__ uint4 post_fwd8(const uint4 &pre, float alpha, unsigned idx, const PostOp &p)
{
    unsigned in[4] = {pre.x, pre.y, pre.z, pre.w}, out[4], rb[2] = {0u, 0u};
    if (p.drop_thr) drop_bits8(idx >> 3, p.drop_seed, rb[0], rb[1]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float a, b;
        unpack2<T>(in[k], a, b);
        float ka = 1.f, kb = 1.f;
        if (p.drop_thr) { ka = drop_factor(rb[k >> 1], 2 * (k & 1), p); kb = drop_factor(rb[k >> 1], 2 * (k & 1) + 1, p); }
        out[k] = repack2(T(), post_fwd1(a, alpha, ka), post_fwd1(b, alpha, kb));
    }
    return make_uint4(out[0], out[1], out[2], out[3]);
}

// kind 2 (y = drop(relu(pre)), `y` in the mask operand): d pre = dy * scale where y > 0 -- no hash, no pre
template <typename T>
__device__ __forceinline__ uint4 post_bwd8_relu(const uint4 &dy, const uint4 &y, float scale)
{
    unsigned g[4] = {dy.x, dy.y, dy.z, dy.w}, q[4] = {y.x, y.y, y.z, y.w}, out[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float ga, gb, ya, yb;
        unpack2<T>(g[k], ga, gb);
        unpack2<T>(q[k], ya, yb);
        out[k] = repack2(T(), ya > 0.f ? ga * scale : 0.f, yb > 0.f ? gb * scale : 0.f);
    }
    return make_uint4(out[0], out[1], out[2], out[3]);
}

template <typename T>
__device__ __forceinline__ uint4 post_bwd8(const uint4 &dy, const uint4 &pre, float alpha, unsigned idx, const PostOp &p, float &dal)
{
    if (p.kind == 2) return post_bwd8_relu<T>(dy, pre, p.drop_scale);
    unsigned g[4] = {dy.x, dy.y, dy.z, dy.w}, q[4] = {pre.x, pre.y, pre.z, pre.w}, out[4], rb[2] = {0u, 0u};
    if (p.drop_thr) drop_bits8(idx >> 3, p.drop_seed, rb[0], rb[1]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float ga, gb, pa, pb;
        unpack2<T>(g[k], ga, gb);
        unpack2<T>(q[k], pa, pb);
        if (p.drop_thr) { ga *= drop_factor(rb[k >> 1], 2 * (k & 1), p); gb *= drop_factor(rb[k >> 1], 2 * (k & 1) + 1, p); }
        out[k] = repack2(T(), post_bwd1(ga, pa, alpha, dal), post_bwd1(gb, pb, alpha, dal));
    }
    return make_uint4(out[0], out[1], out[2], out[3]);
}

// Sum `v` over the lanes of the wave that share `key` and add each sum to slab[key] (LDS, float) -- a tile's rows
// belong to one or two positions of the alpha axis, so this is one or two rounds of a butterfly reduction and one
// LDS atomic each, instead of 64 atomics on the same word.
// sum over the 64 lanes with DPP row shifts / broadcasts (no LDS traffic): lane 63 ends up with the total
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_add(float v)
{
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, false));
}
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v = dpp_add<0x111, 0xf, 0xf>(v);     // row_shr:1
    v = dpp_add<0x112, 0xf, 0xf>(v);     // row_shr:2
    v = dpp_add<0x114, 0xf, 0xe>(v);     // row_shr:4 (banks 1-3)
    v = dpp_add<0x118, 0xf, 0xc>(v);     // row_shr:8 (banks 2-3)
    v = dpp_add<0x142, 0xa, 0xf>(v);     // row_bcast:15 into rows 1, 3
    v = dpp_add<0x143, 0xc, 0xf>(v);     // row_bcast:31 into rows 2, 3
    return v;
}

__device__ __forceinline__ void wave_add_by_key(float v, int key, float *slab, int lane)
{
    const int k0 = __builtin_amdgcn_readfirstlane(key);
    if (__builtin_amdgcn_ballot_w64(key != k0) == 0) {                 // the usual case: one key in the wave
        const float s = wave_sum_to_lane63(v);
        if (lane == 63) atomicAdd(&slab[k0], s);
        return;
    }
    unsigned long long todo = __builtin_amdgcn_ballot_w64(true);
    while (todo) {
        const int first = __builtin_ctzll(todo);
        const int k = __builtin_amdgcn_readlane(key, first);
        const bool mine = key == k;
        const float s = wave_sum_to_lane63(mine ? v : 0.f);
        if (lane == 63) atomicAdd(&slab[k], s);
        todo &= ~__builtin_amdgcn_ballot_w64(mine);
    }
}

}  // namespace qk
