// Small-channel quaternion convolutions on the 16-bit matrix cores: Q, J in {16, 32} gathered / produced channels per component,
// not both 32 -- the 16 -> 16 and 16 -> 32 body layers (forward and backward-data) of a start_filter = 16 TIMIT model
// (/root/reference/models/interspeech_model.py:46-50,109-134; the layer: complexnn/conv.py:288-345).
//
// Why a kernel of its own.  The band kernel (qk_hgemm_bf16mfma.hip) works on 32-channel granules: a 16 -> 16 layer runs it
// zero-padded (PAD form) at 2x the algorithmic MFMA count with 16 MFMAs between two barriers, a tile's prologue and epilogue
// longer than its K loop: 0.16 ms = 22 % of peak (round 5, first half).  Such a layer is small in K x N (960 x 64: 88 GFLOP for
// 184 MB of activations -- on the ridge between the MFMA and the HBM roof), so the kernel has to STREAM:
//   * the WHOLE re-laid-out kernel is resident in LDS (36 - 72 KB: all taps), as ready-made A-operand fragments of
//     v_mfma_f32_16x16x32 (16 filters x 32 K) -- no kernel tiles are staged in the loop, nothing waits for them;
//   * a K step is 32 gathered channels of one inner tap (Q = 32) or 16 channels of TWO inner taps (Q = 16): the B-operand fragment
//     of lane (position l & 15, K group l >> 4) is one 16-byte read of the activation band at row position + tap, so the tap pair is
//     an address, not data movement;
//   * Hamilton structure in registers as everywhere: per K step 4 activation fragments per 16-position group + 4 kernel-part
//     fragments per 16-filter block feed 16 MFMAs; the conj table's negative entries accumulate in a second set (7 tiles);
//   * workgroups are PERSISTENT (one per CU, 8 waves x 16 G positions): a stage = (row tile, outer tap) computes on one band buffer
//     while the next stage's band -- of the same tile or the NEXT one -- arrives in the other by LDS-DMA (source-side swizzle,
//     buffers are objects of their own, a stage ends in vmcnt(0) + barrier: nothing is pending at a merge point);
//   * the accumulator of 16x16x32 holds 4 consecutive filters of one position per lane: 8-byte stores, no transpose.
// Epilogues: bias + relu; forward post-op y = dropout(relu(pre)) (kind 2); backward-data with the producer's mask (chain) and the
// relu + dropout derivative (kind 2).  PReLU post-ops (kind 1) stay on the band kernel's PAD form.
#include "qk_common.h"
#include "qk_postop.h"

namespace qk {
namespace {

typedef __bf16 s16_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 s16_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ floatx4 mfma_s(bf16, const uint4 &a, const uint4 &b, const floatx4 &c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(s16_bf16x8, a), __builtin_bit_cast(s16_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ floatx4 mfma_s(f16, const uint4 &a, const uint4 &b, const floatx4 &c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(s16_f16x8, a), __builtin_bit_cast(s16_f16x8, b), c, 0, 0, 0);
}

typedef unsigned short s16_u2v __attribute__((ext_vector_type(2)));
typedef short s16_s2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned s16_mask2(unsigned v, unsigned m)      // zero the halves whose mask half is <= 0 (mask2 of qk_hgemm_bf16mfma.hip)
{
    const s16_s2v pos = __builtin_elementwise_max(__builtin_bit_cast(s16_s2v, m), (s16_s2v)(0));
    const s16_u2v one = __builtin_elementwise_min(__builtin_bit_cast(s16_u2v, pos), (s16_u2v)(1));
    return v & __builtin_bit_cast(unsigned, (s16_u2v)(one * (s16_u2v)(0xffff)));
}

template <typename T>
__global__ void __launch_bounds__(256)
k_prep_small16(const float *__restrict__ w, T *__restrict__ wq, int Cq, int F, int transposed, int neg_ijk, int kin, int q32, int fb_n,
               long long total)
{
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256)
        prep_small16_write<T>(w, wq, idx, Cq, F, transposed, neg_ijk, kin, q32, fb_n);
    if (blockIdx.x == 0 && threadIdx.x < 128) wq[total + threadIdx.x] = from_f32<T>(0.f);
}

typedef __attribute__((address_space(3))) void s16_lds_void;
constexpr unsigned kOOR16 = 0xF0000000u;

// Residency: the 16 -> 16 form (Q = 16, one filter block: 36 KB of kernel + 2 x 17 KB of band, <= 128 registers) runs TWO workgroups per
// CU -- ablation (tools/ablate_small.py) showed a lone workgroup's phases ADD UP (row decode + DMA issue 40 us, K steps 61 us, epilogue
// 24 us, band wait 13 us of 133): its eight waves do the same thing at the same time; a second, independent workgroup fills them.
template <typename T, bool Q32, int FB, int KIN>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu((!Q32 && FB == 1) ? 4 : 2, (!Q32 && FB == 1) ? 4 : 2)))
k_hconv16_small(const T *__restrict__ in, const uint4 *__restrict__ wq, const float *__restrict__ bias, T *__restrict__ out,
                const GemmGeom g, const int n_tiles, const int w_units, const unsigned out_bytes)
{
    constexpr int G = 1;                            // 16-position groups per wave (128-position tiles)
    constexpr int RW = 16 * G, BM = 8 * RW;         // positions per wave / per tile (128 | 256)
    constexpr int UR = Q32 ? 16 : 8, UQ = UR / 4;   // 16-byte units per band row / per component
    constexpr int KSO = Q32 ? KIN : (KIN + 1) / 2;  // K steps per outer tap
    constexpr int RPP = 512 / UR, NPASS = BM / RPP; // band rows per DMA pass; BM = NPASS full passes + the halo pass of wave 0 (64 / UR rows)
    constexpr int BROWS = BM + 64 / UR;
    constexpr int BAND = BM + KIN - 1;
    static_assert(BM == NPASS * RPP && NPASS <= 4 && KIN - 1 <= 64 / UR, "band = up to four full passes + one halo pass of wave 0");
    static_assert(BROWS * UR * 16 <= 33792, "small16_shape() budgets 33792 bytes per band buffer");
    constexpr unsigned TBL = kSignConj;             // go16 folds the plain table into the kernel
    __shared__ __attribute__((aligned(1024))) uint4 band0[BROWS * UR];
    __shared__ __attribute__((aligned(1024))) uint4 band1[BROWS * UR];
    // the whole kernel: [K step][part][filter block][lane].  A STATIC object (up to three outer taps: small16_shape): with a dynamic
    // `extern __shared__` array hipcc cannot tell its reads from the band buffers the LDS-DMA writes and put a vmcnt(0) in front of
    // the first fragment read of every stage -- right behind the DMA's issue
    constexpr int kMaxOt = 3;
    __shared__ __attribute__((aligned(1024))) uint4 wl[kMaxOt * KSO * 4 * FB * 64];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int WP = g.b_wp;
    const int total_p = g.b_nlines * WP;
    // tiles of this workgroup: every XCD (blockIdx.x % 8) owns a contiguous range of row tiles, its workgroups walk it interleaved --
    // at any time an XCD works on neighbouring tiles, whose bands overlap (the outer taps reach one line up and down): L2 reuse
    const int wpx = gridDim.x >> 3, per_xcd = (n_tiles + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const int tile_end = min(n_tiles, (xcd + 1) * per_xcd);
    int tk = xcd * per_xcd + (blockIdx.x >> 3);
    if (tk >= tile_end) return;

    for (int i = tid; i < w_units; i += 512) wl[i] = wq[i];          // (once per persistent workgroup; the first barrier below covers it)

    // ---- band staging: thread = (row tid / UR of a pass, 16-byte slot tid % UR); the slot HOLDS unit slot ^ swz(row) ------------------
    const int s_row = tid / UR, s_slot = tid % UR;
    const int s_swz = Q32 ? (s_row & 15) : ((s_row >> 1) & 7);       // (RPP and BM are multiples of 16: the same for every pass)
    const int s_unit = s_slot ^ s_swz;
    const unsigned a_thr = (unsigned)((s_unit / UQ) * g.Q + (s_unit % UQ) * 8) * 2u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(in), 0, (int)g.b_in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rmask = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(g.ep_mask ? g.ep_mask : (const void *)out), 0, (int)out_bytes, 0x00020000);
    const unsigned inv_ks1 = (65536u + (unsigned)g.ks[1] - 1u) / (unsigned)g.ks[1];

    int c_off[NPASS + 1];                           // element offsets of this thread's band rows (tap 0) of the tile whose bands are being FETCHED
    unsigned c_om[NPASS + 1];                       // their outer-tap validity masks (a tile's last band is on its way before its last stage starts:
                                                    // the next tile's rows can take the same registers)
    auto decode_tile = [&](int tile, int (&off)[NPASS + 1], unsigned (&om)[NPASS + 1]) -> unsigned {
        const int p0 = tile * BM;
        unsigned tile_ot = 0;
        {
            const int last_p = min(p0 + BAND - 1, total_p - 1);
            const int line_lo = fastdiv(p0, g.dv_mul[0], g.dv_shr[0]);
            const int line_hi = min(fastdiv(last_p, g.dv_mul[0], g.dv_shr[0]), g.b_nlines - 1);
            for (int line = line_lo; line <= line_hi; ++line) {
                const int l2 = fastdiv(line, g.dv_mul[1], g.dv_shr[1]), o1 = line - l2 * g.osp[1];
                const int n = fastdiv(l2, g.dv_mul[2], g.dv_shr[2]), o0 = l2 - n * g.osp[0];
                tile_ot |= outer_tap_mask(o0 * g.pa[0] + g.pc[0], o1 * g.pa[1] + g.pc[1], g);
            }
            tile_ot = __builtin_amdgcn_readfirstlane(tile_ot);
        }
#pragma unroll
        for (int r = 0; r < NPASS + 1; ++r) {
            const int j = r < NPASS ? s_row + r * RPP : BM + s_row;          // (last pass: the halo rows, wave 0 only)
            off[r] = 0; om[r] = 0;
            const int P = p0 + j;
            const int line = fastdiv(P, g.dv_mul[0], g.dv_shr[0]);
            const int col = P - line * WP + g.b_cshift;
            if (j < BAND && line < g.b_nlines && col >= 0 && col < g.isp[2]) {
                const int l2 = fastdiv(line, g.dv_mul[1], g.dv_shr[1]), o1 = line - l2 * g.osp[1];
                const int n = fastdiv(l2, g.dv_mul[2], g.dv_shr[2]), o0 = l2 - n * g.osp[0];
                const int q0 = o0 * g.pa[0] + g.pc[0], q1 = o1 * g.pa[1] + g.pc[1];
                off[r] = n * (int)g.in_sn + q0 * (int)g.in_ss[0] + q1 * (int)g.in_ss[1] + col * (int)g.in_ss[2];
                om[r] = outer_tap_mask(q0, q1, g);
            }
        }
        return tile_ot ? tile_ot : 1u;              // (no valid tap at all: one stage of zero rows, the epilogue still writes bias / activation)
    };
    // band of outer tap `ot` of the tile whose rows are (off, om) -> BUF
#define QK_S_DMA(BUF, OFF, OM, OT) do { \
        const int t0_ = (int)(((unsigned)(OT) * inv_ks1) >> 16), t1_ = (OT) - t0_ * g.ks[1]; \
        const int ad_ = t0_ * g.pb[0] * (int)g.in_ss[0] + t1_ * g.pb[1] * (int)g.in_ss[1]; \
        _Pragma("unroll") for (int r_ = 0; r_ < NPASS + 1; ++r_) { \
            const bool ok_ = ((OM[r_] >> (OT)) & 1u) && !(g.ablate & 16); \
            const unsigned vo_ = ok_ ? (unsigned)(OFF[r_] + ad_) * 2u + a_thr : kOOR16; \
            if (r_ < NPASS) __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (s16_lds_void *)((BUF) + r_ * 512 + wave * 64), 16, (int)vo_, 0, 0, 0); \
            else if (wave == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (s16_lds_void *)((BUF) + NPASS * 512), 16, (int)vo_, 0, 0, 0); \
        } } while (0)

    // ---- fragments ---------------------------------------------------------------------------------------------------------------------
    const int l16 = lane & 15, kg = lane >> 4;
    // B operand (activations): lane = (position l16 of the group, K group kg); Q = 16: K group -> (tap kg >> 1 of the pair, channels 8 (kg & 1) ..),
    // Q = 32: channels 8 kg ..
    const int cu = Q32 ? kg : (kg & 1);             // channel unit inside a component
    floatx4 acc[G][FB][4], accn[G][FB][4];
#define QK_S_ZERO() do { \
        _Pragma("unroll") for (int gq = 0; gq < G; ++gq) \
            _Pragma("unroll") for (int fb = 0; fb < FB; ++fb) \
                _Pragma("unroll") for (int b = 0; b < 4; ++b) \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r) { acc[gq][fb][b][r] = 0.f; accn[gq][fb][b][r] = 0.f; } \
    } while (0)
    // one stage: the KSO K steps of outer tap OT on band buffer BUF
#define QK_S_STAGE(BUF, OT) do { \
        const uint4 *wb_ = wl + (OT) * (KSO * 4 * FB * 64) + lane; \
        _Pragma("unroll") for (int kk = 0; kk < KSO; ++kk) { \
            const int tin_ = Q32 ? kk : 2 * kk + (kg >> 1);                   /* inner tap of this lane's K group */ \
            const int toff_ = tin_ >= KIN ? 0 : (g.b_rev ? KIN - 1 - tin_ : tin_);       /* (phantom tap of an odd pair: zero weights, any row) */ \
            uint4 X[G][4]; \
            _Pragma("unroll") for (int gq = 0; gq < G; ++gq) { \
                const int row_ = wave * RW + gq * 16 + l16 + toff_; \
                const int swz_ = Q32 ? (row_ & 15) : ((row_ >> 1) & 7); \
                _Pragma("unroll") for (int a = 0; a < 4; ++a) X[gq][a] = (BUF)[row_ * UR + ((a * UQ + cu) ^ swz_)]; \
            } \
            _Pragma("unroll") for (int fb = 0; fb < FB; ++fb) { \
                uint4 W[4]; \
                _Pragma("unroll") for (int p = 0; p < 4; ++p) W[p] = wb_[((kk * 4 + p) * FB + fb) * 64]; \
                _Pragma("unroll") for (int gq = 0; gq < G; ++gq) \
                    _Pragma("unroll") for (int a = 0; a < 4; ++a) \
                        _Pragma("unroll") for (int b = 0; b < 4; ++b) { \
                            constexpr unsigned tbl_ = TBL; \
                            if ((tbl_ >> (a * 4 + b)) & 1u) accn[gq][fb][b] = mfma_s(T(), W[a ^ b], X[gq][a], accn[gq][fb][b]); \
                            else acc[gq][fb][b] = mfma_s(T(), W[a ^ b], X[gq][a], acc[gq][fb][b]); \
                        } \
                if (FB > 1) __builtin_amdgcn_sched_barrier(0);          /* (two filter blocks: 112 accumulator registers -- keep the fragments of ONE block live) */ \
            } \
        } } while (0)

    const PostOp psd = resolve_seed(g.post);
    float4 bias_r[FB][4];                           // this lane's four filters of every (filter block, component): kept for the kernel's life
#pragma unroll
    for (int fb = 0; fb < FB; ++fb)
#pragma unroll
        for (int b = 0; b < 4; ++b)
            bias_r[fb][b] = g.has_bias ? *reinterpret_cast<const float4 *>(bias + b * g.J + fb * 16 + 4 * kg) : make_float4(0.f, 0.f, 0.f, 0.f);
    const bool post_fwd_relu = g.post.kind == 2 && g.post_fwd != 0;           // y = dropout(relu(pre)), one output tensor
    const bool relu_bwd = g.post.kind == 2 && g.ep_mask != nullptr && !g.post_fwd;   // d pre = dy / (1 - rate) where y > 0
    typedef unsigned u2x __attribute__((ext_vector_type(2)));
    // ---- the stage pipeline --------------------------------------------------------------------------------------------------------------
    unsigned cur_mask = decode_tile(tk, c_off, c_om);
    {
        const int ot0 = __builtin_ctz(cur_mask);
        QK_S_DMA(band0, c_off, c_om, ot0);
    }
    QK_S_ZERO();
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    bool running = true;
    while (running) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {          // two stages per trip: the band buffer of a stage is a compile-time OBJECT
            const int ot = __builtin_ctz(cur_mask);
            const unsigned rem = cur_mask & (cur_mask - 1u);
            const bool last_of_tile = rem == 0u;
            const int tn = tk + wpx;
            const bool have_next = !last_of_tile || tn < tile_end;
            unsigned nxt_mask = rem;
            if (last_of_tile && have_next) nxt_mask = decode_tile(tn, c_off, c_om);
            if (have_next) {
                const int otn = __builtin_ctz(nxt_mask);
                if (par == 0) QK_S_DMA(band1, c_off, c_om, otn); else QK_S_DMA(band0, c_off, c_om, otn);
            }
            if (!(g.ablate & 4)) { if (par == 0) QK_S_STAGE(band0, ot); else QK_S_STAGE(band1, ot); }
            if (last_of_tile) {
                if (!(g.ablate & 32)) {
                    const int p0 = tk * BM;
                    unsigned erow_[G];
                    bool ok_[G];
                    u2x mk_[G][FB][4];
#pragma unroll
        for (int gq = 0; gq < G; ++gq) {                 // every mask load of the tile first: one round trip, behind the band in flight
            const int P = p0 + wave * RW + gq * 16 + l16;
            const int line = fastdiv(P, g.dv_mul[0], g.dv_shr[0]);
            const int u = P - line * WP;
            ok_[gq] = line < g.b_nlines && u < g.osp[2];
            erow_[gq] = (unsigned)((line * g.osp[2] + u) * (int)g.out_ss);      // element offset of the position's row
            if (g.ep_mask) {
#pragma unroll
                for (int fb = 0; fb < FB; ++fb)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        mk_[gq][fb][b] = __builtin_amdgcn_raw_buffer_load_b64(rmask, ok_[gq] ? (int)((erow_[gq] + (unsigned)(b * g.J + fb * 16 + 4 * kg)) * 2u) : (int)kOOR16, 0, 0);
            }
        }
#pragma unroll
        for (int gq = 0; gq < G; ++gq) {
            const bool ok = ok_[gq];
            const unsigned erow = erow_[gq];
#pragma unroll
            for (int fb = 0; fb < FB; ++fb)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int ch = b * g.J + fb * 16 + 4 * kg;                           // this lane's four filters of component b
                    const unsigned eidx = erow + (unsigned)ch;
                    const unsigned voff = (ok && !(g.ablate & 8)) ? eidx * 2u : kOOR16;
                    float v[4];
                    const float bv[4] = {bias_r[fb][b].x, bias_r[fb][b].y, bias_r[fb][b].z, bias_r[fb][b].w};
                    constexpr unsigned tbl_ = TBL;
                    constexpr unsigned col_neg = (tbl_ | tbl_ >> 4 | tbl_ >> 8 | tbl_ >> 12) & 0xfu;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = acc[gq][fb][b][r] + bv[r];
                        if ((col_neg >> b) & 1u) v[r] -= accn[gq][fb][b][r];
                        if (g.relu) v[r] = v[r] > 0.f ? v[r] : 0.f;
                    }
                    unsigned d0 = repack2(T(), v[0], v[1]), d1 = repack2(T(), v[2], v[3]);
                    if (g.ep_mask) {
                        const u2x mk = mk_[gq][fb][b];
                        if (relu_bwd) {             // (as post_bwd8_relu: on the 16-bit values, scale in fp32)
                            float ga, gb, ya, yb;
                            unpack2<T>(d0, ga, gb); unpack2<T>(mk.x, ya, yb);
                            d0 = repack2(T(), ya > 0.f ? ga * psd.drop_scale : 0.f, yb > 0.f ? gb * psd.drop_scale : 0.f);
                            unpack2<T>(d1, ga, gb); unpack2<T>(mk.y, ya, yb);
                            d1 = repack2(T(), ya > 0.f ? ga * psd.drop_scale : 0.f, yb > 0.f ? gb * psd.drop_scale : 0.f);
                        } else { d0 = s16_mask2(d0, mk.x); d1 = s16_mask2(d1, mk.y); }
                    }
                    if (post_fwd_relu) {            // (as post_fwd8 on the half unit this lane holds: elements eidx .. eidx + 3 of unit eidx >> 3)
                        unsigned lo = 0u, hi = 0u;
                        if (psd.drop_thr) drop_bits8(eidx >> 3, psd.drop_seed, lo, hi);
                        const unsigned bits = (eidx & 4u) ? hi : lo;
                        float a0, a1, a2, a3;
                        unpack2<T>(d0, a0, a1); unpack2<T>(d1, a2, a3);
                        float k0 = 1.f, k1 = 1.f, k2 = 1.f, k3 = 1.f;
                        if (psd.drop_thr) { k0 = drop_factor(bits, 0, psd); k1 = drop_factor(bits, 1, psd); k2 = drop_factor(bits, 2, psd); k3 = drop_factor(bits, 3, psd); }
                        d0 = repack2(T(), post_fwd1(a0, 0.f, k0), post_fwd1(a1, 0.f, k1));
                        d1 = repack2(T(), post_fwd1(a2, 0.f, k2), post_fwd1(a3, 0.f, k3));
                    }
                    u2x st; st.x = d0; st.y = d1;
                    __builtin_amdgcn_raw_buffer_store_b64(st, rout, (int)voff, 0, 0);
                }
        }
                }
                QK_S_ZERO();
                if (have_next) tk = tn;
            }
            cur_mask = nxt_mask;
            // the next band has landed, this one has been read.  A tile's output stores are YOUNGER than the band's DMA (the counter is in
            // order): they stay in flight across the barrier -- vmcnt(0) here waited for their write acknowledgements, ~2 us per tile
            constexpr int NST = G * FB * 4;
            if (last_of_tile) __builtin_amdgcn_s_waitcnt((NST & 15) | (7 << 4) | (0 << 8) | ((NST >> 4) << 14));
            else __builtin_amdgcn_s_waitcnt((7 << 4) | (0 << 8));
            __builtin_amdgcn_s_barrier();
            if (!have_next) { running = false; break; }
        }
    }
#undef QK_S_DMA
#undef QK_S_STAGE
#undef QK_S_ZERO
}

template <typename T, bool Q32, int FB, int KIN>
int run_small16(const T *in, const uint4 *wq, const float *bias, T *out, const GemmGeom &bg, const Small16 &s, hipStream_t stream)
{
    constexpr int BM = 128;
    constexpr int WG_PER_CU = (!Q32 && FB == 1) ? 2 : 1;
    const int n_tiles = (int)(((long long)bg.b_nlines * bg.b_wp + BM - 1) / BM);
    int blocks = device_cu_count() * WG_PER_CU / 8 * 8;          // persistent workgroups: what is resident at once
    if (blocks < 8) blocks = 8;
    while (blocks > 8 && (blocks >> 3) > (n_tiles + 7) / 8) blocks -= 8;
    const unsigned out_bytes = (unsigned)((long long)bg.M * bg.out_ss * 2);
    hipLaunchKernelGGL((k_hconv16_small<T, Q32, FB, KIN>), dim3(blocks), dim3(512), 0, stream, in, wq, bias, out, bg, n_tiles,
                       (int)(s.w_bytes / 16), out_bytes);
    return hipGetLastError() == hipSuccess ? 1 : QK_ERR_LAUNCH;
}

template <typename T>
int go_small16(const void *in, const void *wq, const float *bias, void *out, const GemmGeom &bg, const Small16 &s, hipStream_t stream)
{
    const T *ip = (const T *)in;
    const uint4 *wp = (const uint4 *)wq;
    T *op = (T *)out;
    const bool k5 = s.kin == 5;
    if (s.q32) return k5 ? run_small16<T, true, 1, 5>(ip, wp, bias, op, bg, s, stream) : run_small16<T, true, 1, 3>(ip, wp, bias, op, bg, s, stream);
    if (s.fb == 2) return k5 ? run_small16<T, false, 2, 5>(ip, wp, bias, op, bg, s, stream) : run_small16<T, false, 2, 3>(ip, wp, bias, op, bg, s, stream);
    return k5 ? run_small16<T, false, 1, 5>(ip, wp, bias, op, bg, s, stream) : run_small16<T, false, 1, 3>(ip, wp, bias, op, bg, s, stream);
}

}  // namespace

int launch_prep_small16(int dtype, const float *w, void *wq, int Cq, int F, int transposed, int neg_ijk, const Small16 &s, hipStream_t stream)
{
    const long long total = (long long)s.w_bytes / 2;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    if (dtype == QK_BF16) hipLaunchKernelGGL((k_prep_small16<bf16>), dim3(blocks), dim3(256), 0, stream, w, (bf16 *)wq, Cq, F, transposed, neg_ijk, s.kin, s.q32, s.fb, total);
    else if (dtype == QK_F16) hipLaunchKernelGGL((k_prep_small16<f16>), dim3(blocks), dim3(256), 0, stream, w, (f16 *)wq, Cq, F, transposed, neg_ijk, s.kin, s.q32, s.fb, total);
    else return QK_ERR_INVALID_ARG;
    return hipGetLastError() == hipSuccess ? 0 : QK_ERR_LAUNCH;
}

// 1: took the call; 0: the device refused (caller falls back); < 0: error.  bg = the band geometry of small16_shape (fastdiv fields set by the caller).
int launch_hconv16_small(int dtype, const void *in, const void *wq, const float *bias, void *out, const GemmGeom &bg, const Small16 &s, hipStream_t stream)
{
    if (dtype == QK_BF16) return go_small16<bf16>(in, wq, bias, out, bg, s, stream);
    if (dtype == QK_F16) return go_small16<f16>(in, wq, bias, out, bg, s, stream);
    return QK_ERR_INVALID_ARG;
}

}  // namespace qk
