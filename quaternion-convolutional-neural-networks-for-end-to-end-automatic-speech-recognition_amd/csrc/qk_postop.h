// Post-op of a quaternion layer inside the TIMIT model: keras PReLU followed by Dropout
// (models/interspeech_model.py:99-101,117-121 of the reference: `O = PReLU(shared_axes=[1,0])(O); O = Dropout(d.dropout)(O)`)
//
//     y = drop(prelu(pre)),   prelu(v) = v > 0 ? v : alpha * v,   drop(v) = keep ? v / (1 - rate) : 0
//
// fused into the epilogue of the kernel that produces `pre` (forward: both `pre` and `y` are written) and, in the
// backward, into the epilogue of the NEXT layer's backward-data kernel, whose output IS the gradient w.r.t. y:
//
//     g = drop'(dy_out),   d pre = pre > 0 ? g : pre < 0 ? alpha * g : 0,   d alpha += sum over (pre < 0) of g * pre
//
// alpha is either one scalar or one slope per position along ONE spatial axis of the layer output (what Keras'
// `shared_axes=[1, 0]` yields for a channels_first (C, F, T) tensor: (1, F, 1), see layers.PReLU).  The dropout mask
// is not stored: it is a counter-based hash of (seed, flat element index of y in its channels-last buffer), evaluated
// again in the backward.  One hash serves two neighbouring elements (16 bits each).
#pragma once
#include "qk_common.h"

namespace qk {

__device__ __forceinline__ unsigned drop_hash(unsigned pair_index, unsigned seed)
{
    unsigned h = (pair_index ^ seed) * 0x9E3779B1u;
    h ^= h >> 15; h *= 0x85EBCA77u;
    h ^= h >> 13; h *= 0xC2B2AE3Du;
    h ^= h >> 16;
    return h;
}

// scale factor of element `e` (0 / 1) of the pair whose hash is h: 0 when dropped
__device__ __forceinline__ float drop_factor(unsigned h, int e, const PostOp &p)
{
    const unsigned v = e ? (h >> 16) : (h & 0xffffu);
    return v >= p.drop_thr ? p.drop_scale : 0.f;
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
    return (v > 0.f ? v : alpha * v) * keep;
}
// d pre and the slope-gradient term of one element; g = dy * keep
__device__ __forceinline__ float post_bwd1(float g, float pre, float alpha, float &dal)
{
    dal += pre < 0.f ? g * pre : 0.f;
    return pre > 0.f ? g : (pre < 0.f ? alpha * g : 0.f);
}

// 8 consecutive 16-bit elements starting at flat index `idx` (a multiple of 8)
template <typename T>
__device__ __forceinline__ uint4 post_fwd8(const uint4 &pre, float alpha, unsigned idx, const PostOp &p)
{
    unsigned in[4] = {pre.x, pre.y, pre.z, pre.w}, out[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float a, b;
        unpack2<T>(in[k], a, b);
        float ka = 1.f, kb = 1.f;
        if (p.drop_thr) { const unsigned h = drop_hash((idx >> 1) + k, p.drop_seed); ka = drop_factor(h, 0, p); kb = drop_factor(h, 1, p); }
        out[k] = repack2(T(), post_fwd1(a, alpha, ka), post_fwd1(b, alpha, kb));
    }
    return make_uint4(out[0], out[1], out[2], out[3]);
}

template <typename T>
__device__ __forceinline__ uint4 post_bwd8(const uint4 &dy, const uint4 &pre, float alpha, unsigned idx, const PostOp &p, float &dal)
{
    unsigned g[4] = {dy.x, dy.y, dy.z, dy.w}, q[4] = {pre.x, pre.y, pre.z, pre.w}, out[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float ga, gb, pa, pb;
        unpack2<T>(g[k], ga, gb);
        unpack2<T>(q[k], pa, pb);
        if (p.drop_thr) { const unsigned h = drop_hash((idx >> 1) + k, p.drop_seed); ga *= drop_factor(h, 0, p); gb *= drop_factor(h, 1, p); }
        out[k] = repack2(T(), post_bwd1(ga, pa, alpha, dal), post_bwd1(gb, pb, alpha, dal));
    }
    return make_uint4(out[0], out[1], out[2], out[3]);
}

// Sum `v` over the lanes of the wave that share `key` and add each sum to slab[key] (LDS, float) -- a tile's rows
// belong to one or two positions of the alpha axis, so this is one or two rounds of a butterfly reduction and one
// LDS atomic each, instead of 64 atomics on the same word.
__device__ __forceinline__ void wave_add_by_key(float v, int key, float *slab, int lane)
{
    unsigned long long todo = __builtin_amdgcn_ballot_w64(true);
    while (todo) {
        const int first = __builtin_ctzll(todo);
        const int k = __builtin_amdgcn_readlane(key, first);
        const bool mine = key == k;
        float s = mine ? v : 0.f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == first) atomicAdd(&slab[k], s);
        todo &= ~__builtin_amdgcn_ballot_w64(mine);
    }
}

}  // namespace qk
