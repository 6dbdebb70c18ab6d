// Backward-weight (+ bias gradient) on the 16-bit-input matrix cores, gfx950.
//
//   dW[t, c, p, f] = sum_{a^b=p} sgn(a,b) * sum_m x_a[pos(m,t), c] * dy_b[m, f]
//
// The reduction runs over rows m, which is the NON-contiguous axis of both operands
// (activations are [row][channel]).  v_mfma_f32_32x32x16 wants 8 consecutive reduction elements
// per lane, so the X / dY tiles stay in LDS exactly as they lie in HBM and the fragments are
// formed with the gfx950 LDS transpose read ds_read_b64_tr_b16: inside a 16-lane group, lane i
// receives element (i & 3) of the 8-byte words addressed by lanes 4j + (i >> 2), j = 0..3
// (measured, tools/tr_probe.hip).  Addressing lane L at (row m0 + L/4, channels c0 + 4*(L%4)..+3)
// therefore hands lane i channel c0 + i for rows m0..m0+3: two reads give the 8-deep fragment.
//
// Block = (tap, BC input channels x BF filters, split of M); it accumulates the EXPANDED gradient
// tile (4*BC rows x 4*BF columns) in MFMA accumulators -- wave tile 64 x TN*32 -- and folds the 16
// (a,b) blocks onto the 4 compact parts through LDS in four race-free phases before one atomic
// pass to HBM, as the fp32 kernel does (qk_hgemm_f32mfma.inc).  K step = 64 rows of M, two LDS
// buffers + a register stage, one barrier per step; the staging stores / loads are issued between
// the MFMAs of the step (pinned with sched_barrier), not in a phase of their own.  Row pitches carry 64 bytes of padding so the
// four rows a transpose read touches per 32-lane half fall on different bank quarters.
#include "qk_common.h"

namespace qk {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));

__device__ __forceinline__ floatx16 mfma16w(bf16, const v8s &a, const v8s &b, const floatx16 &c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ floatx16 mfma16w(f16, const v8s &a, const v8s &b, const floatx16 &c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

typedef __attribute__((address_space(3))) v4s lds_v4s;

// 8-deep fragment for this lane's channel: rows [m, m+4) and [m+4, m+8) of the tile at `base`
__device__ __forceinline__ v8s tr_frag(const char *base, int pitch)
{
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(base));
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(base + 4 * pitch));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Staging goes through buffer resources: `buffer_load_dwordx4 v, voffset, s[rsrc], soffset offen`
// takes a 32-bit per-lane byte offset plus a wave-uniform one, so a load costs no address VALU at all
// (flat 64-bit pointers cost ~4 VALU per load here), and an offset beyond the resource's extent
// returns zeros -- padding rows and rows past the split need no select afterwards.  The kernel was
// VALU-bound on exactly that work: 291 VALU per wave and K step next to 32 MFMAs (PMC).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOutOfRange = 0xF0000000u;          // > every extent try_wgrad_16 admits
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void buf_store16(const uint4 &d, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    u32x4 v; v.x = d.x; v.y = d.y; v.z = d.z; v.w = d.w;
    // (soffset stays 0: gfx950 store-data hazard, see buf_store16b in qk_hgemm_bf16mfma.hip)
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)(voff + soff), 0, 0);
}

// keep the 16-bit halves of v whose mask half is non-zero.  The mask is this layer's RELU output
// (include/qk.h), i.e. >= +0, so "non-zero" == "> 0": v_pk_min_u16, v_pk_mul_lo_u16, v_and.
// (Two tempting shortcuts do not survive: `v * min(m, 1)` is turned into compare + select + permute
// by hipcc, 5.5 VALU per dword; the same two instructions as inline asm stored a stale first dword
// on some lanes -- the hazard recogniser does not see through the asm.)
typedef unsigned short u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned relu_keep2(unsigned v, unsigned m)
{
    const u2v one = __builtin_elementwise_min(__builtin_bit_cast(u2v, m), (u2v)(1));
    return v & __builtin_bit_cast(unsigned, (u2v)(one * (u2v)(0xffff)));
}

// TG = taps per block.  With 32 input channels a (tap, all channels) block is only 128 rows tall;
// two taps side by side ("virtual channels" [tap member][component][channel]) restore the 256-row
// tile: the dY / Y tiles are then staged once for two taps (half the L2 traffic and half the mask
// VALU per MFMA).  A group's missing last tap is staged as zeros.
template <typename T, int WR, int WC, int TN, int TG, bool MASK>
__global__ void __launch_bounds__(WR * WC * 64) __attribute__((amdgpu_waves_per_eu(WR * WC / 4, 2)))
k_wgrad16(const T *__restrict__ x, const T *__restrict__ dy, const T *__restrict__ ymask,
          float *__restrict__ dw, float *__restrict__ dbias, const WgradGeom g)
{
    constexpr int NW = WR * WC, NTHR = NW * 64;
    constexpr int BC = WR * 16;                 // "virtual" quaternion input channels per block (rows = 4*BC)
    constexpr int BCQ = BC / TG;                // real input channels per block and tap
    constexpr int BF = WC * TN * 8;             // quaternion filters per block      (cols = 4*BF)
    constexpr int KM = 64;                      // rows of M per K step
    constexpr int XROW = 4 * BC * 2 + 64;       // bytes per tile row (64 B pad: see header)
    constexpr int DROW = 4 * BF * 2 + 64;
    constexpr int BUF = KM * (XROW + DROW);
    constexpr int FOLD = BC * 4 * BF * 4;
    static_assert(2 * BUF >= FOLD, "fold slab reuses the tile buffers");
    static_assert(BCQ % 32 == 0, "a 32-row MFMA tile must lie inside one (tap member, component)");
    __shared__ __attribute__((aligned(16))) char lds[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    // XCD-aware order (workgroup b runs on XCD b % 8, speed only): the (tap, channel-chunk) blocks of
    // one split of M sit next to each other on one XCD, so the dY / X rows they all read are served
    // by that XCD's L2 (without this the 30 blocks of a split re-fetch them from HBM: 7x over-fetch).
    const int nfc = g.F / BF;
    const int n_grp = (g.taps + TG - 1) / TG;   // tap groups
    const int n_inner = n_grp * (g.Cq / BCQ) * nfc;
    const int n_tiles = n_inner * g.n_splits;
    const int per_xcd = (n_tiles + 7) / 8;
    const int tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (tile >= n_tiles) return;
    const int split = tile / n_inner;
    const int inner = tile - split * n_inner;
    const int t = inner % n_grp;                // tap group of this block
    const int chunk = inner / n_grp;
    const int cchunk = chunk / nfc;
    const int fchunk = chunk - cchunk * nfc;
    const int c0 = cchunk * BCQ, f0 = fchunk * BF;
    int tp0[TG], tp1[TG], tp2[TG];              // kernel position of each member tap
    bool tp_ok[TG];
#pragma unroll
    for (int mbr = 0; mbr < TG; ++mbr) {
        const int tm = t * TG + mbr;
        tp_ok[mbr] = tm < g.taps;
        tp2[mbr] = tm % g.ks[2];
        const int tt = tm / g.ks[2];
        tp1[mbr] = tt % g.ks[1];
        tp0[mbr] = tt / g.ks[1];
    }
    const int m_begin = split * g.m_per_split;
    const int m_end = min(g.M, m_begin + g.m_per_split);
    // dbias and the masked-dY side output need every dY row exactly once.  The tap-group blocks of one
    // (split, filter chunk) all stage the same dY tiles, so they take turns: group block t owns the K
    // steps with step % n_grp == t.  (One owner block would run ~30 % longer than its peers and, with
    // one workgroup per CU, set the kernel time.)
    // (deterministic mode: tap group 0 takes every turn -- one addition per bias column)
    const int n_turns = g.deterministic ? (t == 0 ? 1 : (1 << 30)) : n_grp;
    const bool bias_blk = g.want_dbias && cchunk == 0;
    int bias_turn = g.deterministic ? (t == 0 ? 0 : (1 << 29)) : t;      // 0 => this K step is ours

    // ---- staging: NTHR/64 threads per row, 16-byte units ------------------------------------
    constexpr int TPROW = NTHR / KM;
    constexpr int UXR = BC / 2, UDR = BF / 2;                // 16-byte units per X / dY row
    constexpr int UX = (UXR + TPROW - 1) / TPROW, UD = (UDR + TPROW - 1) / TPROW;
    static_assert(UXR % TPROW == 0 && UDR % TPROW == 0, "units per thread");
    const int s_row = tid / TPROW, s_sub = tid % TPROW;
    uint4 xr[UX], dr[UD], mr[MASK ? UD : 1];
    // Per-lane byte offsets into the buffer resources: the register stage holds tile j+1 while tile j
    // is multiplied; `vd_cur` is the dY offset of the tile in the registers (masked-dY store), nvx / nvd
    // those of the tile the next loads fetch.  Rows outside the tensor or the split get kOutOfRange.
    constexpr int UC = BCQ / 8, UF = BF / 8;                 // 16-byte units per component block
    constexpr int UM = 4 * UC;                               // units per tap member of an X row
    static_assert(TPROW % UC == 0 && UM % TPROW == 0 && TPROW % UF == 0,
                  "unit -> (member, component, channel group) must split into a per-thread and a per-unit part");
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, g.x_bytes), rdy = make_rsrc(dy, g.dy_bytes);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(MASK ? ymask : dy, g.dy_bytes);
    const __amdgpu_buffer_rsrc_t rdym = make_rsrc(g.dym ? g.dym : dy, g.dym ? g.dy_bytes : 0u);
    // X unit q = s_sub + u*TPROW of a row: member q / UM, component (q % UM) / UC, channel group q % UC
    const unsigned x_thr = (unsigned)((s_sub / UC) * g.Cq + c0 + (s_sub % UC) * 8) * 2u;
    const unsigned d_thr = (unsigned)((s_sub / UF) * g.F + f0 + (s_sub % UF) * 8) * 2u;
    const unsigned x_cstep = (unsigned)g.Cq * 2u;            // one component further (wave-uniform)
    const unsigned d_ustep = (unsigned)((TPROW / UF) * g.F) * 2u;
    unsigned nvx[TG], nvd = kOutOfRange, vd_cur = kOutOfRange;
#pragma unroll
    for (int mbr = 0; mbr < TG; ++mbr) nvx[mbr] = kOutOfRange;
    // position of this thread's row, advanced by KM rows per decode (no divisions in the loop)
    int r_n, r_o0, r_o1, r_o2;
    {
        int s = m_begin + s_row;
        r_o2 = s % g.osp[2]; s /= g.osp[2];
        r_o1 = s % g.osp[1]; s /= g.osp[1];
        r_o0 = s % g.osp[0];
        r_n = s / g.osp[0];
    }
    const bool dym_blk = MASK && g.dym != nullptr && cchunk == 0;
    int dym_turn = g.deterministic ? (t == 0 ? 0 : (1 << 29)) : t;   // same rotation for the staged tiles

    auto decode_next = [&](int mb) {
        const int m = mb + s_row;
        const bool d_in = m < m_end;
        const int xb = r_n * (int)g.x_sn;
#pragma unroll
        for (int mbr = 0; mbr < TG; ++mbr) {
            const int i0 = r_o0 * g.pa[0] + tp0[mbr] * g.pb[0] + g.pc[0];
            const int i1 = r_o1 * g.pa[1] + tp1[mbr] * g.pb[1] + g.pc[1];
            const int i2 = r_o2 * g.pa[2] + tp2[mbr] * g.pb[2] + g.pc[2];
            const bool x_in = d_in && tp_ok[mbr] && i0 >= 0 && i0 < g.isp[0] && i1 >= 0 && i1 < g.isp[1] && i2 >= 0 && i2 < g.isp[2];
            const int xo = xb + i0 * (int)g.x_ss[0] + i1 * (int)g.x_ss[1] + i2 * (int)g.x_ss[2];
            nvx[mbr] = x_in ? (unsigned)xo * 2u + x_thr : kOutOfRange;
        }
        nvd = d_in ? (unsigned)(m * (int)g.dy_ss) * 2u + d_thr : kOutOfRange;
        // next call: KM rows further.  Carry-propagate when the innermost extent is long (images);
        // re-decode with divisions when it is short (dense layers, 1-D convolutions: osp[2] == 1)
        if (g.osp[2] >= KM) {
            r_o2 += KM;
            const bool c2 = r_o2 >= g.osp[2];
            r_o2 -= c2 ? g.osp[2] : 0;
            r_o1 += c2 ? 1 : 0;
            const bool c1 = r_o1 == g.osp[1];
            r_o1 = c1 ? 0 : r_o1;
            r_o0 += c1 ? 1 : 0;
            const bool c0_ = r_o0 == g.osp[0];
            r_o0 = c0_ ? 0 : r_o0;
            r_n += c0_ ? 1 : 0;
        } else {
            int s = m + KM;
            r_o2 = s % g.osp[2]; s /= g.osp[2];
            r_o1 = s % g.osp[1]; s /= g.osp[1];
            r_o0 = s % g.osp[0];
            r_n = s / g.osp[0];
        }
    };
    auto commit_next = [&]() { vd_cur = nvd; };
    // unit u of the staged pair (X units first, then dY): one 16-byte load / one 16-byte LDS store
    auto load_unit = [&](int u) {
        if (u < UX) {
            xr[u] = buf_load16(rx, nvx[(u * TPROW) / UM], (((u * TPROW) % UM) / UC) * x_cstep);
        } else {
            const int i = u - UX;
            dr[i] = buf_load16(rdy, nvd, i * d_ustep);
            if constexpr (MASK) mr[i] = buf_load16(ry, nvd, i * d_ustep);
        }
    };
    auto store_unit = [&](int u, int buf, bool write_dym) {
        if (u < UX) {
            char *xs = lds + buf * BUF + s_row * XROW;
            *reinterpret_cast<uint4 *>(xs + (s_sub + u * TPROW) * 16) = xr[u];
        } else {
            const int i = u - UX, q = s_sub + i * TPROW;
            char *ds = lds + buf * BUF + KM * XROW + s_row * DROW;
            uint4 v = dr[i];
            if constexpr (MASK)
                v = make_uint4(relu_keep2(v.x, mr[i].x), relu_keep2(v.y, mr[i].y), relu_keep2(v.z, mr[i].z),
                               relu_keep2(v.w, mr[i].w));
            *reinterpret_cast<uint4 *>(ds + q * 16) = v;
            if constexpr (MASK) {
                if (write_dym) {
                    buf_store16(v, rdym, vd_cur, i * d_ustep);                // rows past the split: dropped
                    // A 16-byte buffer store reads its data registers for many cycles after issue.  hipcc
                    // pads a following VALU write of them with 2 wait states (none when soffset is an
                    // SGPR); on gfx950 that corrupted the first dwords on some lanes (the registers are
                    // reused as address temporaries right away).  16 wait states, on 1/taps of the steps.
                    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
                }
            }
        }
    };
    auto dym_now = [&]() {
        const bool w = dym_blk && dym_turn == 0;
        dym_turn = dym_turn == 0 ? n_turns - 1 : dym_turn - 1;
        return w;
    };

    floatx16 acc[2][TN];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < TN; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][ct][r] = 0.f;
    float dbacc = 0.f;

    // transpose-read addressing of this lane (see header): 16-lane group g16 covers channels
    // 16*g16 .. +15 of the 32-channel tile, half kh the second 8 rows of the 16-deep step
    const int Lg = lane & 15, g16 = (lane >> 4) & 1, kh = lane >> 5;
    const int fr_row = 8 * kh + (Lg >> 2);
    const int fr_ch = 16 * g16 + 4 * (Lg & 3);
    const int a_off = fr_row * XROW + ((wr * 2) * 32 + fr_ch) * 2;                 // + rt*64 bytes
    const int b_off = KM * XROW + fr_row * DROW + ((wc * TN) * 32 + fr_ch) * 2;      // + ct*64 bytes

    constexpr int NU = UX + UD;                      // staged 16-byte units per thread and K step
    constexpr int KS = KM / 16;                      // 16-deep MFMA steps per K step
    constexpr int NMF = 2 * TN;                      // MFMAs per 16-deep step
    static_assert((NU + KS - 1) / KS * 2 <= NMF, "two issue slots per staged unit");
    const int iters = (m_end - m_begin + KM - 1) / KM;
    if (iters > 0) {
        decode_next(m_begin);
#pragma unroll
        for (int u = 0; u < NU; ++u) load_unit(u);
        commit_next();
        {
            const bool w = dym_now();
#pragma unroll
            for (int u = 0; u < NU; ++u) store_unit(u, 0, w);
        }
        decode_next(m_begin + KM);
#pragma unroll
        for (int u = 0; u < NU; ++u) load_unit(u);
        commit_next();
        __syncthreads();
        for (int it = 0; it < iters; ++it) {
            // Tile `it` is multiplied out of buffer it & 1 while tile it + 1 moves from the register
            // stage into the other buffer and tile it + 2 is fetched into the freed registers.  The
            // staging instructions are pinned BETWEEN the MFMAs (one store or one load per slot), so
            // they issue in the shadow of the matrix pipe instead of in a phase of their own.
            const char *tb = lds + (it & 1) * BUF;
            const int nb = (it + 1) & 1;
            const bool bias_now = bias_blk && bias_turn == 0;
            bias_turn = bias_turn == 0 ? n_turns - 1 : bias_turn - 1;
            if (bias_now && tid < 4 * BF) {
                const T *col = reinterpret_cast<const T *>(tb + KM * XROW) + tid;
#pragma unroll 8
                for (int mm = 0; mm < KM; ++mm) dbacc += to_f32(col[mm * (DROW / 2)]);
            }
            const bool w = dym_now();
            decode_next(m_begin + (it + 2) * KM);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                // (prefetching the next step's fragments during these MFMAs was measured: +2 % on the
                // linear variant, but the RELU variant then spills; not worth a second code path)
                v8s A[2], B[TN];
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) A[rt] = tr_frag(tb + a_off + ks * 16 * XROW + rt * 64, XROW);
#pragma unroll
                for (int ct = 0; ct < TN; ++ct) B[ct] = tr_frag(tb + b_off + ks * 16 * DROW + ct * 64, DROW);
                __builtin_amdgcn_sched_barrier(0);
                const int u_lo = ks * NU / KS, u_hi = (ks + 1) * NU / KS;
#pragma unroll
                for (int j = 0; j < NMF; ++j) {
                    const int rt = j / TN, ct = j % TN;
                    acc[rt][ct] = mfma16w(T(), A[rt], B[ct], acc[rt][ct]);
                    const int u = u_lo + j / 2;
                    if (u < u_hi) {
                        if (j % 2 == 0) store_unit(u, nb, w);
                        else load_unit(u);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            commit_next();
            __syncthreads();
        }
    }

    // ---- fold the 16 expanded blocks onto the 4 compact parts, then one atomic pass ----------
    // Phase ph deposits the rows of gathered component a = ph onto part p = ph ^ b.  Waves that hold
    // the same a differ in their output components b (disjoint p), or in their channels cc, so the
    // plain LDS read-modify-write is race free; phase 0 (p = b) initialises every slab entry.
    if ((g.ablate & 1) && acc[0][0][0] != 123.456f) return;          // (ablate 1: profiling, no fold / atomics)
    float *slab = reinterpret_cast<float *>(lds);
    const int lr = lane & 31;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int rowbase = (wr * 2 + rt) * 32;                    // rows: [member][component][BCQ]
            if ((rowbase / BCQ) % 4 != ph) continue;                  // wave-uniform: BCQ is 32 or 64
            const int mbr = rowbase / (4 * BCQ);
#pragma unroll
            for (int ct = 0; ct < TN; ++ct) {
                const int col = (wc * TN + ct) * 32 + lr;
                const int b = col / BF, ff = col % BF;
                const int p = ph ^ b;
                const bool neg = (g.sign_tbl >> (ph * 4 + b)) & 1u;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cc = (rowbase + mfma32_row(r, lane)) % BCQ;
                    const float v = neg ? -acc[rt][ct][r] : acc[rt][ct][r];
                    float *dst = &slab[((mbr * BCQ + cc) * 4 + p) * BF + ff];
                    *dst = ph == 0 ? v : *dst + v;
                }
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < BC * 4 * BF; e += NTHR) {
        const int ff = e % BF;
        const int p = (e / BF) & 3;
        const int cv = e / (4 * BF);                                    // virtual channel: member * BCQ + cc
        const int tm = t * TG + cv / BCQ;
        if (tm < g.taps && !(g.ablate & 2))
            atomicAdd(dw + ((g.w_ch_major ? (c0 + cv % BCQ) * g.taps + tm : tm * g.Cq + c0 + cv % BCQ) * 4 + p) * g.F + f0 + ff, slab[e]);
    }
    if (bias_blk && tid < 4 * BF) {
        const int b = tid / BF, ff = tid % BF;
        atomicAdd(dbias + b * g.F + f0 + ff, dbacc);
    }
}

template <typename T, int WR, int WC, int TN, int TG>
int run_wgrad16(const T *x, const T *dy, const T *ymask, float *dw, float *dbias, WgradGeom g,
                hipStream_t stream)
{
    constexpr int BCQ = WR * 16 / TG, BF = WC * TN * 8, KM = 64;
    const int ncc = g.Cq / BCQ, nfc = g.F / BF;
    const long long other = (long long)ncc * nfc * ((g.taps + TG - 1) / TG);
    const long long max_splits = ((long long)g.M + KM - 1) / KM;
    // Split M so that the grid fills a whole number of residency rounds: with 144 KB of LDS there is
    // one workgroup per CU, and e.g. 780 equal tiles on 256 CUs take four rounds, the last 5 % full.
    // resident workgroups per CU follow from the kernel's static LDS (160 KB per CU); the CU count is asked of
    // the current device on every call (multi-GPU processes, no cached state)
    constexpr int kLdsBytes = 2 * KM * ((4 * WR * 16 * 2 + 64) + (4 * WC * TN * 8 * 2 + 64));
    constexpr int per_cu = (160 * 1024) / kLdsBytes > 0 ? (160 * 1024) / kLdsBytes : 1;
    const int slots = device_cu_count() * (per_cu > 2 ? 2 : per_cu);        // 8-wave blocks, 2 waves per SIMD: <= 2
    const long long kEpilogueSteps = 16;                   // fold + atomics, in K-step equivalents
    long long splits = 1, best_cost = -1;
    for (int r = 1; r <= 4; ++r) {
        long long sp = (long long)r * slots / other;
        sp = sp < 1 ? 1 : (sp > max_splits ? max_splits : sp);
        long long mps_r = (g.M + sp - 1) / sp;
        mps_r = (mps_r + KM - 1) / KM * KM;
        sp = (g.M + mps_r - 1) / mps_r;
        const long long rounds = (sp * other + slots - 1) / slots;
        const long long cost = rounds * (mps_r / KM + kEpilogueSteps);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; splits = sp; }
    }
    g.deterministic = (debug_flags() & kDbgDeterministic) ? 1 : 0;
    if (g.deterministic) splits = 1;
    long long mps = (g.M + splits - 1) / splits;
    mps = (mps + KM - 1) / KM * KM;
    splits = (g.M + mps - 1) / mps;
    g.m_per_split = (int)mps;
    g.n_splits = (int)splits;
    g.x_bytes = (unsigned)((long long)g.batch * g.x_sn * 2);
    g.dy_bytes = (unsigned)((long long)g.M * g.dy_ss * 2);
    g.ablate = debug_ablate();
    const long long n_tiles = splits * other;
    dim3 grid((unsigned)((n_tiles + 7) / 8 * 8), 1, 1);      // padded to the 8 XCDs (see the tile remap)
    if (g.has_mask)
        hipLaunchKernelGGL((k_wgrad16<T, WR, WC, TN, TG, true>), grid, dim3(WR * WC * 64), 0, stream, x, dy, ymask, dw, dbias, g);
    else
        hipLaunchKernelGGL((k_wgrad16<T, WR, WC, TN, TG, false>), grid, dim3(WR * WC * 64), 0, stream, x, dy, ymask, dw, dbias, g);
    return hipGetLastError() == hipSuccess ? 1 : QK_ERR_LAUNCH;
}

template <typename T>
int go_wgrad16(const void *x, const void *dy, const void *ymask, float *dw, float *dbias, const WgradGeom &g,
               hipStream_t stream)
{
    const T *xp = (const T *)x, *dp = (const T *)dy, *yp = (const T *)ymask;
    const bool one_tap = (debug_flags() & kDbgWgradOneTap) != 0;      // tuning aid: the TG = 1 tilings only
    if (g.Cq % 64 == 0) {
        if (g.F % 64 == 0) return run_wgrad16<T, 4, 2, 4, 1>(xp, dp, yp, dw, dbias, g, stream);
        return run_wgrad16<T, 4, 2, 2, 1>(xp, dp, yp, dw, dbias, g, stream);
    }
    // 32-channel chunks: two taps per block make the same 256-row tiles (dY staged once for both)
    if (g.taps > 1 && !one_tap) {
        if (g.F % 64 == 0) return run_wgrad16<T, 4, 2, 4, 2>(xp, dp, yp, dw, dbias, g, stream);
        return run_wgrad16<T, 4, 2, 2, 2>(xp, dp, yp, dw, dbias, g, stream);
    }
    // single tap (dense layers, 1x1 convolutions): 8 waves of 64 x 64 rather than 4 waves of 64 x 128
    if (g.F % 64 == 0) return run_wgrad16<T, 2, 4, 2, 1>(xp, dp, yp, dw, dbias, g, stream);
    return run_wgrad16<T, 2, 2, 2, 1>(xp, dp, yp, dw, dbias, g, stream);
}

}  // namespace

// Returns 1 when the 16-bit MFMA path took the call, 0 when the shape is outside its fast path,
// < 0 on error.  dw / dbias must already be zeroed (the kernel accumulates atomically).
int try_wgrad_16(int dtype, const void *x, const void *dy, const void *ymask, float *dw, float *dbias,
                 const WgradGeom &g, hipStream_t stream)
{
    if (dtype != QK_BF16 && dtype != QK_F16) return 0;
    if (g.x_sc != 1 || g.dy_sc != 1) return 0;                         // channels_last buffers only
    if (g.Cq % 32 != 0 || g.F % 32 != 0) return 0;
    // buffer-resource addressing: 32-bit byte offsets, kOutOfRange must stay beyond every extent
    if ((long long)g.batch * g.x_sn * 2 >= 0xF0000000ll || (long long)g.M * g.dy_ss * 2 >= 0xF0000000ll) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(ymask)) & 15) return 0;
    if (debug_flags() & kDbgNoMfma16) return 0;
    note_path(QK_PATH_MFMA16);
    if (dtype == QK_BF16) return go_wgrad16<bf16>(x, dy, ymask, dw, dbias, g, stream);
    return go_wgrad16<f16>(x, dy, ymask, dw, dbias, g, stream);
}

}  // namespace qk
