// Backward-weight on the 16-bit-input matrix cores, BAND form: one block computes the kernel gradient of ALL
// taps along the innermost spatial axis (KIN = 3 or 5) for one (outer tap, channel chunk, filter chunk).
//
//   dW[t0, t1, t, c, p, f] = sum_{a^b=p} sgn(a,b) * sum_P x_a[P + t, c] * dy_b[P, f]        t = 0 .. KIN-1
//
// k_wgrad16 (qk_wgrad_bf16mfma.hip) gives every tap its own block, so the X and dY tiles of a row range are staged
// once PER TAP: 15 x (|x| + |dy|) through L2 and LDS for a (3,5) kernel -- its 32-channel instantiations are
// LDS-bound (33 % of the MFMA peak) and the 64-channel one moves 1.3x the algorithmic HBM bytes.  Here the
// reduction index P runs over PADDED lines of the innermost axis (out extent + KIN - 1 positions per line,
// exactly as in k_hgemm16_band): then "input position of (P, inner tap t)" is simply "band row P + t", one staged
// X band of KM + KIN - 1 rows serves all KIN taps, and the dY tile is staged once for them.  Padded positions
// carry a zero dY row (buffer loads past the extent return zeros), so they contribute nothing.
//
// Block tile = KIN taps x (4 components x CQB channels) rows x (4 components x BF filters) columns of the EXPANDED
// gradient = KIN x RTT x CT tiles of 32 x 32, ten per wave (160 accumulator registers):
//   RTT = 2 (F % 64 == 0): CQB = 16 channels, BF = 64 filters;  wave = (row tile rt, component b = cg), 2 col tiles
//   RTT = 4 (F % 32 == 0): CQB = 32 channels, BF = 32 filters;  wave = (component a = rt, column half),  2 col tiles
//   RTT = 2, CTW = 1 (masked, five taps, Cq % 32 != 0): CQB = 16, BF = 32, ONE col tile per wave (five accumulator tiles): the
//   64-filter masked form needs eight staged dY / mask units per thread and spilled 4 - 32 registers (round-3 verdict)
// Per 16-deep step a wave reads KIN A fragments (one per tap: the same band, shifted by one row) and 2 B fragments
// with ds_read_b64_tr_b16 and issues 2 KIN MFMAs.  Fold of the 16 (a,b) blocks onto the 4 compact parts, bias
// gradient, masked-dY side output and the XCD-aware block order are those of k_wgrad16.
#include "qk_common.h"

namespace qk {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));

__device__ __forceinline__ floatx16 mfma16b(bf16, const v8s &a, const v8s &b, const floatx16 &c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ floatx16 mfma16b(f16, const v8s &a, const v8s &b, const floatx16 &c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

typedef __attribute__((address_space(3))) v4s lds_v4s;
__device__ __forceinline__ v8s tr_frag8(const char *base, int pitch)
{
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(base));
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(base + 4 * pitch));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOOR = 0xF0000000u;                  // > every extent try_wgrad_band_16 admits
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 bload16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void bstore16(const uint4 &d, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    u32x4 v; v.x = d.x; v.y = d.y; v.z = d.z; v.w = d.w;
    // (soffset stays 0: gfx950 store-data hazard, see buf_store16b in qk_hgemm_bf16mfma.hip)
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)(voff + soff), 0, 0);
}
typedef unsigned short u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned keep2(unsigned v, unsigned m)       // see relu_keep2 in qk_wgrad_bf16mfma.hip
{
    const u2v one = __builtin_elementwise_min(__builtin_bit_cast(u2v, m), (u2v)(1));
    return v & __builtin_bit_cast(unsigned, (u2v)(one * (u2v)(0xffff)));
}

template <typename T, int RTT, int KIN, bool MASK, int CTW>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_wgrad16_band(const T *__restrict__ x, const T *__restrict__ dy, const T *__restrict__ ymask,
               float *__restrict__ dw, float *__restrict__ dbias, const WgradGeom g)
{
    static_assert(RTT == 2 || RTT == 4, "row tiles per tap");
    static_assert(KIN == 3 || KIN == 5, "inner taps");
    static_assert(CTW == 1 || CTW == 2, "column tiles per wave");
    constexpr int NTHR = 512, WCG = 8 / RTT;
    constexpr int CQB = RTT * 8, BF = WCG * CTW * 8;          // channels / filters per block
    constexpr int KM = 64;                                     // positions per K step
    constexpr int XROW = 4 * CQB * 2 + 64, DROW = 4 * BF * 2 + 64;   // (+64 B: transpose reads stay conflict free)
    constexpr int XBUF = (KM + KIN - 1) * XROW, DBUF = KM * DROW, BUF = XBUF + DBUF;
    constexpr int FOLD = KIN * CQB * 4 * BF * 4;
    static_assert(2 * BUF >= FOLD, "fold slab reuses the tile buffers");
    __shared__ __attribute__((aligned(16))) char lds[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = wave / WCG, cg = wave % WCG;
    // ---- which block (XCD-aware: the blocks of one split of the positions sit on one XCD and share x / dy in its L2)
    const int n_ot = g.ks[0] * g.ks[1];
    const int ncc = g.Cq / CQB, nfc = g.F / BF;
    const int n_inner = n_ot * ncc * nfc;
    const int n_tiles = n_inner * g.n_splits;
    const int per_xcd = (n_tiles + 7) / 8;
    const int tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (tile >= n_tiles) return;
    const int split = tile / n_inner;
    const int inner = tile - split * n_inner;
    const int ot = inner % n_ot;                    // outer tap (t0 * ks1 + t1)
    const int chunk = inner / n_ot;
    const int cchunk = chunk / nfc, fchunk = chunk - cchunk * nfc;
    const int c0 = cchunk * CQB, f0 = fchunk * BF;
    const int t0 = ot / g.ks[1], t1 = ot - t0 * g.ks[1];
    const int WP = g.b_wp, W = g.osp[2];
    const int p_begin = split * g.m_per_split;
    const int p_end = min(g.b_nlines * WP, p_begin + g.m_per_split);
    // the blocks that stage the same dY tiles -- (outer tap, channel chunk) of one (split, filter chunk) -- take turns
    // at the bias gradient and at the masked-dY side output (one owner would set the kernel time)
    // (deterministic mode: ONE owner -- the block of outer tap 0 / channel chunk 0 -- takes every turn, so each bias
    //  column receives a single addition; the others never do)
    const bool share0 = ot * ncc + cchunk == 0;
    const int n_share = g.deterministic ? (share0 ? 1 : (1 << 30)) : n_ot * ncc;
    const bool bias_blk = g.want_dbias != 0, dym_blk = MASK && g.dym != nullptr;
    int turn = g.deterministic ? (share0 ? 0 : (1 << 29)) : ot * ncc + cchunk;      // 0 => this K step is ours

    // ---- staging: 8 threads per position, 16-byte units -------------------------------------------------------
    constexpr int UC = CQB / 8, UF = BF / 8;        // units per component block of an X / dY row
    constexpr int UX = (4 * UC) / 8, UD = (4 * UF) / 8;       // units per thread
    static_assert(UX >= 1 && UD >= 1, "units per thread");
    const int s_row = tid >> 3, s_sub = tid & 7;
    const __amdgpu_buffer_rsrc_t rx = rsrc_of(x, g.x_bytes), rdy = rsrc_of(dy, g.dy_bytes);
    const __amdgpu_buffer_rsrc_t ry = rsrc_of(MASK ? ymask : dy, g.dy_bytes);
    const __amdgpu_buffer_rsrc_t rdym = rsrc_of(g.dym ? g.dym : dy, g.dym ? g.dy_bytes : 0u);
    // unit q = s_sub + 8 u of a row: component q / UC (UF), channel group q % UC (UF)
    const unsigned x_thr = (unsigned)((s_sub / UC) * g.Cq + c0 + (s_sub % UC) * 8) * 2u;
    const unsigned d_thr = (unsigned)((s_sub / UF) * g.F + f0 + (s_sub % UF) * 8) * 2u;
    const unsigned x_ustep = (unsigned)((8 / UC) * g.Cq) * 2u, d_ustep = (unsigned)((8 / UF) * g.F) * 2u;
    uint4 xr[UX], xh[UX], dr[UD], mr[MASK ? UD : 1];
    // position of this thread's row: (line, u) with line = (n * osp0 + o0) * osp1 + o1; advanced by KM per decode.
    // Round 4: the offsets themselves are carried along (xo2 / dyo2: byte offsets of the row's x / dY units, exact modulo 2^32
    // whenever the row is inside the tensor) -- an advance is additions of wave-uniform constants selected by the carries, no
    // multiplications and no branches, so its pieces can sit between the MFMAs of the K step instead of in front of them
    // (the decode was 11 % of the kernel by ablation: ~60 VALU with six v_mul_lo / v_mad_u64 and nested exec-mask branches
    // that every wave of the CU ran right behind the barrier, with the matrix pipe idle).
    int r_line, r_u, r_o0, r_o1, r_i0, r_i1;
    unsigned xo2, dyo2;
    const int W_ = g.osp[2];
    {
        const int P = p_begin + s_row;
        r_line = P / WP; r_u = P - r_line * WP;
        int l = r_line;
        r_o1 = l % g.osp[1]; l /= g.osp[1];
        r_o0 = l % g.osp[0];
        const int r_n = l / g.osp[0];
        r_i0 = r_o0 * g.pa[0] + t0 * g.pb[0] + g.pc[0];
        r_i1 = r_o1 * g.pa[1] + t1 * g.pb[1] + g.pc[1];
        xo2 = (unsigned)(r_n * (int)g.x_sn + r_i0 * (int)g.x_ss[0] + r_i1 * (int)g.x_ss[1] + (r_u + g.b_cshift) * (int)g.x_ss[2]) * 2u + x_thr;
        dyo2 = (unsigned)((r_line * W_ + r_u) * (int)g.dy_ss) * 2u + d_thr;
    }
    // wave-uniform steps: KM positions along a line; line -> next line; last line of axis 1 -> next row of axis 0; -> next sample
    const unsigned cXA = (unsigned)(KM * (int)g.x_ss[2]) * 2u, cDA = (unsigned)(KM * (int)g.dy_ss) * 2u;
    const unsigned cXL = (unsigned)((int)g.x_ss[1] * g.pa[1] - WP * (int)g.x_ss[2]) * 2u, cDL = (unsigned)((W_ - WP) * (int)g.dy_ss) * 2u;
    const unsigned cX1 = (unsigned)((int)g.x_ss[0] * g.pa[0] - g.osp[1] * g.pa[1] * (int)g.x_ss[1]) * 2u;
    const unsigned cX0 = (unsigned)((int)g.x_sn - g.osp[0] * g.pa[0] * (int)g.x_ss[0]) * 2u;
    const int cI1 = g.osp[1] * g.pa[1], cI0 = g.osp[0] * g.pa[0];
    constexpr bool INTERLEAVE = !(MASK && KIN == 5);   // (the masked 5-tap forms have no register to spare for the longer live ranges)
    const bool fast_lines = WP >= KM;                // at most one line end per advance (every image wider than 60 positions)
    // offsets of the row this thread loads next (vx: band row of the tile, vd: its dY row) and of the one after
    // (vx_n: also the HALO row of the tile, band row KM + s_row, for the threads with s_row < KIN - 1)
    unsigned vx = kOOR, vd = kOOR, vx_n = kOOR, vd_n = kOOR, vd_cur = kOOR;
    int p_next = p_begin;                            // position base of the next decode
    bool carry = false, carry1 = false;
    // decode = piece 0 (offsets of the current row) + pieces 1 .. 3 (advance by KM positions)
    auto decode_piece = [&](int k) {
        if (k == 0) {
            const bool in_img = r_line < g.b_nlines;
            const bool x_in = in_img && (unsigned)(r_u + g.b_cshift) < (unsigned)g.isp[2] && (unsigned)r_i0 < (unsigned)g.isp[0]
                              && (unsigned)r_i1 < (unsigned)g.isp[1];
            vx_n = x_in ? xo2 : kOOR;
            const bool d_in = in_img && r_u < W_ && p_next + s_row < p_end;
            vd_n = d_in ? dyo2 : kOOR;
            p_next += KM;
        } else if (k == 1) {
            r_u += KM; xo2 += cXA; dyo2 += cDA;
            carry = r_u >= WP;
            r_u -= carry ? WP : 0; r_line += carry ? 1 : 0; r_o1 += carry ? 1 : 0; r_i1 += carry ? g.pa[1] : 0;
            xo2 += carry ? cXL : 0u; dyo2 += carry ? cDL : 0u;
        } else if (k == 2) {
            carry1 = carry && r_o1 == g.osp[1];
            r_o1 = carry1 ? 0 : r_o1; r_i1 -= carry1 ? cI1 : 0; r_o0 += carry1 ? 1 : 0; r_i0 += carry1 ? g.pa[0] : 0;
            xo2 += carry1 ? cX1 : 0u;
        } else {
            const bool carry0 = carry1 && r_o0 == g.osp[0];
            r_o0 = carry0 ? 0 : r_o0; r_i0 -= carry0 ? cI0 : 0;
            xo2 += carry0 ? cX0 : 0u;
        }
    };
    auto decode_short_lines = [&]() {                // lines shorter than a K step: several line ends per advance -- decode afresh
        decode_piece(0);
        const int P = p_next + s_row;
        r_line = P / WP; r_u = P - r_line * WP;
        int l = r_line;
        r_o1 = l % g.osp[1]; l /= g.osp[1];
        r_o0 = l % g.osp[0];
        const int r_n = l / g.osp[0];
        r_i0 = r_o0 * g.pa[0] + t0 * g.pb[0] + g.pc[0];
        r_i1 = r_o1 * g.pa[1] + t1 * g.pb[1] + g.pc[1];
        xo2 = (unsigned)(r_n * (int)g.x_sn + r_i0 * (int)g.x_ss[0] + r_i1 * (int)g.x_ss[1] + (r_u + g.b_cshift) * (int)g.x_ss[2]) * 2u + x_thr;
        dyo2 = (unsigned)((r_line * W_ + r_u) * (int)g.dy_ss) * 2u + d_thr;
    };
    auto decode = [&]() {
        if (fast_lines) { decode_piece(0); decode_piece(1); decode_piece(2); decode_piece(3); }
        else decode_short_lines();
    };
    auto shift = [&]() { vx = vx_n; vd = vd_n; };
    const bool halo_thr = s_row < KIN - 1;
    constexpr int NU = 2 * UX + UD;                  // staged units per thread and K step: X, X halo, dY
    auto load_unit = [&](int u) {
        if (u < UX) xr[u] = bload16(rx, vx, u * x_ustep);
        else if (u < 2 * UX) { if (wave == 0) xh[u - UX] = bload16(rx, halo_thr ? vx_n : kOOR, (u - UX) * x_ustep); }
        else {
            const int i = u - 2 * UX;
            dr[i] = bload16(rdy, vd, i * d_ustep);
            if constexpr (MASK) mr[i] = bload16(ry, vd, i * d_ustep);
        }
    };
    auto store_unit = [&](int u, int buf, bool write_dym) {
        char *b = lds + buf * BUF;
        if (u < UX) *reinterpret_cast<uint4 *>(b + s_row * XROW + (s_sub + u * 8) * 16) = xr[u];
        else if (u < 2 * UX) {
            if (wave == 0 && halo_thr) *reinterpret_cast<uint4 *>(b + (KM + s_row) * XROW + (s_sub + (u - UX) * 8) * 16) = xh[u - UX];
        } else {
            const int i = u - 2 * UX;
            uint4 v = dr[i];
            if constexpr (MASK) v = make_uint4(keep2(v.x, mr[i].x), keep2(v.y, mr[i].y), keep2(v.z, mr[i].z), keep2(v.w, mr[i].w));
            *reinterpret_cast<uint4 *>(b + XBUF + s_row * DROW + (s_sub + i * 8) * 16) = v;
            if constexpr (MASK) {
                if (write_dym) {
                    bstore16(v, rdym, vd_cur, i * d_ustep);
                    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");      // store-data hazard, see k_wgrad16
                }
            }
        }
    };

    floatx16 acc[KIN][CTW];
#pragma unroll
    for (int t = 0; t < KIN; ++t)
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][ct][r] = 0.f;
    float dbacc = 0.f;

    const int Lg = lane & 15, g16 = (lane >> 4) & 1, kh = lane >> 5;
    const int fr_row = 8 * kh + (Lg >> 2);
    const int fr_ch = 16 * g16 + 4 * (Lg & 3);
    const int a_off = fr_row * XROW + (rt * 32 + fr_ch) * 2;                          // + (ks*16 + t) rows
    const int b_off = XBUF + fr_row * DROW + ((cg * CTW) * 32 + fr_ch) * 2;           // + ks*16 rows, + ct*64 bytes

    constexpr int KS = KM / 16, NMF = KIN * CTW;
    static_assert((NU + KS - 1) / KS * 2 <= NMF, "two issue slots per staged unit");
    const int iters = (p_end - p_begin + KM - 1) / KM;
    if (iters > 0) {
        decode(); shift(); decode();                 // vx/vd: tile 0, vx_n: tile 1 (= halo rows of tile 0)
#pragma unroll
        for (int u = 0; u < NU; ++u) load_unit(u);
        vd_cur = vd;
        {
            const bool mine = turn == 0;
#pragma unroll
            for (int u = 0; u < NU; ++u) store_unit(u, 0, dym_blk && mine);
        }
        shift(); decode();
#pragma unroll
        for (int u = 0; u < NU; ++u) load_unit(u);
        vd_cur = vd;
        __syncthreads();
        for (int it = 0; it < iters; ++it) {
            const char *tb = lds + (it & 1) * BUF;
            const int nb = (it + 1) & 1;
            const bool mine = turn == 0;              // this tile's dY rows are ours (bias / masked dY): tile `it`
            turn = turn == 0 ? n_share - 1 : turn - 1;
            const bool mine_next = turn == 0;         // ... and the one moving into LDS now: tile it + 1
            if (bias_blk && mine && tid < 4 * BF) {
                const T *col = reinterpret_cast<const T *>(tb + XBUF) + tid;
#pragma unroll 8
                for (int mm = 0; mm < KM; ++mm) dbacc += to_f32(col[mm * (DROW / 2)]);
            }
            shift();
            if (!INTERLEAVE || !fast_lines) decode();        // (else: the four pieces ride between the MFMAs of ks = 0)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                v8s A[KIN], B[CTW];
                // The KIN tap fragments of a lane are windows [t, t + 8) of the SAME 12 consecutive band positions of its
                // channel (tap t pairs x[P + t] with dy[P]): three transposing reads fetch the 12 positions once and the
                // windows are cut out of the six dwords in registers -- even taps are plain register sub-ranges, odd taps
                // one v_alignbit_b32 per dword -- instead of 2 KIN reads of overlapping rows (the kernel was LDS-bound:
                // 14 ds_read_b64_tr_b16 per 10 MFMAs; now 7).
                {
                    const char *ab = tb + a_off + (ks * 16) * XROW;
                    const v4s q0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(ab));
                    const v4s q1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(ab + 4 * XROW));
                    const v4s q2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(ab + 8 * XROW));
                    typedef unsigned u2 __attribute__((ext_vector_type(2)));
                    const u2 e0 = __builtin_bit_cast(u2, q0), e1 = __builtin_bit_cast(u2, q1), e2 = __builtin_bit_cast(u2, q2);
                    const unsigned d[6] = {e0.x, e0.y, e1.x, e1.y, e2.x, e2.y};
#pragma unroll
                    for (int t = 0; t < KIN; ++t) {
                        u32x4 f;
                        if (t % 2 == 0) { f.x = d[t / 2]; f.y = d[t / 2 + 1]; f.z = d[t / 2 + 2]; f.w = d[t / 2 + 3]; }
                        else {
                            const int o = t / 2;
                            f.x = __builtin_amdgcn_alignbit(d[o + 1], d[o], 16); f.y = __builtin_amdgcn_alignbit(d[o + 2], d[o + 1], 16);
                            f.z = __builtin_amdgcn_alignbit(d[o + 3], d[o + 2], 16); f.w = __builtin_amdgcn_alignbit(d[o + 4], d[o + 3], 16);
                        }
                        A[t] = __builtin_bit_cast(v8s, f);
                    }
                }
#pragma unroll
                for (int ct = 0; ct < CTW; ++ct) B[ct] = tr_frag8(tb + b_off + ks * 16 * DROW + ct * 64, DROW);
                __builtin_amdgcn_sched_barrier(0);
                const int u_lo = ks * NU / KS, u_hi = (ks + 1) * NU / KS;
#pragma unroll
                for (int j = 0; j < NMF; ++j) {
                    const int t = j / CTW, ct = j % CTW;
                    acc[t][ct] = mfma16b(T(), A[t], B[ct], acc[t][ct]);
                    if (INTERLEAVE && ks == 0 && j >= 1 && j <= 4 && fast_lines) decode_piece(j - 1);      // vx_n / vd_n: first used in ks = 1
                    const int u = u_lo + j / 2;
                    if (u < u_hi) {
                        if (j % 2 == 0) store_unit(u, nb, dym_blk && mine_next);
                        else load_unit(u);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            vd_cur = vd;
            __syncthreads();
        }
    }

    // ---- fold the 16 expanded blocks onto the 4 compact parts, then one atomic pass ---------------------------
    if ((g.ablate & 1) && acc[0][0][0] != 123.456f) return;
    float *slab = reinterpret_cast<float *>(lds);     // [tap][channel][part][filter]
    const int lr = lane & 31;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {                  // phase = gathered component a
        const bool active = RTT == 2 ? rt == (ph >> 1) : rt == ph;
        if (active) {
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct) {
                const int col = (cg * CTW + ct) * 32 + lr;
                const int b = col / BF, ff = col % BF;
                const int p = ph ^ b;
                const bool neg = (g.sign_tbl >> (ph * 4 + b)) & 1u;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (RTT == 2 && (r >> 3) != (ph & 1)) continue;      // this register's rows belong to the other component
                    const int row = mfma32_row(r, lane);
                    const int cc = RTT == 2 ? (row & 15) : row;
#pragma unroll
                    for (int t = 0; t < KIN; ++t) {
                        const float v = neg ? -acc[t][ct][r] : acc[t][ct][r];
                        float *dst = &slab[((t * CQB + cc) * 4 + p) * BF + ff];
                        *dst = ph == 0 ? v : *dst + v;
                    }
                }
            }
        }
        __syncthreads();
    }
    const int tap0 = ot * KIN;                        // compact tap index of inner tap 0
    for (int e = tid; e < KIN * CQB * 4 * BF; e += NTHR) {
        const int ff = e % BF;
        const int p = (e / BF) & 3;
        const int cv = e / (4 * BF);                  // tap * CQB + channel
        const int t = cv / CQB, cc = cv - t * CQB;
        if (!(g.ablate & 2))
            atomicAdd(dw + (((tap0 + t) * g.Cq + c0 + cc) * 4 + p) * g.F + f0 + ff, slab[e]);
    }
    if (bias_blk && tid < 4 * BF) {
        const int b = tid / BF, ff = tid % BF;
        atomicAdd(dbias + b * g.F + f0 + ff, dbacc);
    }
}


// ---------------------------------------------------------------------------------------
// Round 5: the LINEAR form (no relu mask to apply: what a chain of layers runs) staged by LDS-DMA, three tile buffers.
//
// What bound k_wgrad16_band (65 % MFMA-busy on all-zero operands, 29 % of the wave cycles parked, 3.2 non-MFMA VALU per MFMA):
//   * one barrier per 64-position K step with nothing in flight across it -- behind the barrier all eight waves issue their
//     first fragment reads and the matrix pipe idles for an LDS round trip, 544 times per block;
//   * tiles travelled global -> 24 staging registers -> ds_write_b128; the per-thread row decode ran on the VALU in every wave;
//     the tap windows were cut out of registers (v_alignbit / moves of misaligned register tuples: 12 VALU per 10 MFMAs).
// Here:
//   * X and dY tiles go L2 -> LDS by `buffer_load_dwordx4 ... lds`.  Every tile is a set of PLANES [row][128 bytes] (8 DMA
//     lanes fill one row of a plane: thread = (row tid >> 3, unit tid & 7) as before), the two 64-byte blocks of a row swapped
//     for rows with (row >> 1) & 1 on the SOURCE side: the four rows a transposing read touches per 32 lanes then sit on the
//     four 16-bank groups (pad-free, conflict-free for every tap offset);
//   * THREE tile buffers (objects of their own; the loop is unrolled by three so that every buffer is a compile-time object --
//     hipcc orders a ds_read behind every pending LDS-DMA into the same object): tile it + 2 is asked for in the second half of
//     step it and is waited for in the middle of step it + 1 (a whole K step to arrive), in front of the ONE barrier of a step,
//     which also says that tile it - 1's buffer is free.  The fragments of a sub-step are read during the sub-step before it
//     (two fragment sets), across the step boundary too: nothing drains at the barrier;
//   * the tap windows are LDS addresses (tap t = the same transposing read t rows further down: 2 KIN + 4 reads per sub-step,
//     immediate offsets, no VALU); rows are decoded from wave-uniform LINE state kept in scalar registers (a K step touches two
//     padded lines at most) -- per thread one compare and a few selects; the bias gradient comes from v_dot2c on the dY
//     fragments the MFMAs use anyway (no LDS column reads).
// ---------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot_ones(bf16, unsigned w, float c)
{
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w), __builtin_bit_cast(bf16x2_t, 0x3F803F80u), c, false);
}
__device__ __forceinline__ float dot_ones(f16, unsigned w, float c)
{
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, w), __builtin_bit_cast(f16x2_t, 0x3C003C00u), c, false);
}
__device__ __forceinline__ v8s tr_pair(const lds_char *p)                 // rows r .. r + 3 and r + 4 .. r + 7 of a 128-byte-pitch plane
{
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(p));
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s *)(p + 512));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// PAD (RTT = 4 form only): Cq and / or F are multiples of 16 but not of 32 (start_filter = 16 models) -- the units of channels
// >= Cq / filters >= F are out-of-range DMA lanes (zeros in LDS), gradient entries beyond the real extents are not written.
// CTW = 1 (RTT = 2 only): ONE column tile per wave -- 16 channels x 32 filters per block, KIN accumulator tiles per wave -- for
// Cq = 16 (mod 32) layers whose filter count is not a multiple of 64 (16 -> 16, 16 -> 32): the channel side is exact, the filter
// side pads to 32 at most (the 32 x 32 form would pad both sides).
template <typename T, int RTT, int KIN, bool PAD = false, int CTW = 2>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_wgrad16_band3(const T *__restrict__ x, const T *__restrict__ dy, float *__restrict__ dw, float *__restrict__ dbias, const WgradGeom g)
{
    static_assert(!PAD || RTT == 4 || CTW == 1, "padded channel counts: the 32 x 32 block form, or 16 x 32 with an exact channel side");
    static_assert(CTW == 2 || (CTW == 1 && RTT == 2), "column tiles per wave");
    static_assert(RTT == 2 || RTT == 4, "row tiles per tap");
    static_assert(KIN == 3 || KIN == 5, "inner taps");
    constexpr int NTHR = 512, WCG = 8 / RTT;
    constexpr int CQB = RTT * 8, BF = WCG * CTW * 8;          // channels / filters per block
    constexpr int KM = 64, KS = KM / 16, NMF = KIN * CTW;
    constexpr int NPX = (4 * CQB * 2) / 128, NPD = (4 * BF * 2) / 128;     // 128-byte planes of an X / dY row
    constexpr int XROWS = KM + 8;                             // + the halo rows (KIN - 1 <= 8: one DMA of wave 0)
    constexpr int XPL = XROWS * 128, DPL = KM * 128;          // bytes of one plane
    constexpr int DOFF = NPX * XPL;                           // dY planes behind the X planes
    constexpr int BUFB = NPX * XPL + NPD * DPL;
    constexpr int TAPB = CQB * 4 * BF * 4;                    // fold slab of one tap
    static_assert(2 * TAPB <= BUFB, "two taps of the fold slab per tile buffer");
    static_assert((KIN + 1) / 2 <= 3, "three tile buffers");
    __shared__ __attribute__((aligned(1024))) char lds0[BUFB];
    __shared__ __attribute__((aligned(1024))) char lds1[BUFB];
    __shared__ __attribute__((aligned(1024))) char lds2[BUFB];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = wave / WCG, cg = wave % WCG;
    // ---- which block (as k_wgrad16_band)
    const int n_ot = g.ks[0] * g.ks[1];
    const int ncc = (g.Cq + CQB - 1) / CQB, nfc = (g.F + BF - 1) / BF;       // (PAD: the last chunk may be half real)
    const int n_inner = n_ot * ncc * nfc;
    const int n_tiles = n_inner * g.n_splits;
    const int per_xcd = (n_tiles + 7) / 8;
    const int tile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (tile >= n_tiles) return;
    const int split = tile / n_inner;
    const int inner = tile - split * n_inner;
    const int ot = inner % n_ot;
    const int chunk = inner / n_ot;
    const int cchunk = chunk / nfc, fchunk = chunk - cchunk * nfc;
    const int c0 = cchunk * CQB, f0 = fchunk * BF;
    const int t0 = ot / g.ks[1], t1 = ot - t0 * g.ks[1];
    const int WP = g.b_wp, W = g.osp[2];
    const int p_begin = split * g.m_per_split;
    const int p_end = min(g.b_nlines * WP, p_begin + g.m_per_split);
    const bool share0 = ot * ncc + cchunk == 0;
    const int n_share = g.deterministic ? (share0 ? 1 : (1 << 30)) : n_ot * ncc;
    const bool bias_wave = g.want_dbias != 0 && rt == 0;      // the waves of row tile 0 hold every column tile of the block once
    int turn = g.deterministic ? (share0 ? 0 : (1 << 29)) : ot * ncc + cchunk;      // 0 => this K step's bias sums are ours

    // ---- staging: thread = (row s_row, 16-byte unit s_sub) of a plane; the unit it FETCHES is s_sub with the row's block swap
    constexpr int UC = CQB / 8, UF = BF / 8;        // units per component block of an X / dY row
    const int s_row = tid >> 3, s_sub = tid & 7;
    const int s_src = s_sub ^ (((s_row >> 1) & 1) << 2);
    const __amdgpu_buffer_rsrc_t rx = rsrc_of(x, g.x_bytes), rdy = rsrc_of(dy, g.dy_bytes);
    const unsigned x_thr = (unsigned)((s_src / UC) * g.Cq + c0 + (s_src % UC) * 8) * 2u;
    const unsigned d_thr = (unsigned)((s_src / UF) * g.F + f0 + (s_src % UF) * 8) * 2u;
    const unsigned x_pstep = (unsigned)((8 / UC) * g.Cq) * 2u, d_pstep = (unsigned)((8 / UF) * g.F) * 2u;     // plane -> plane (wave-uniform)
    // PAD: this thread's 8-channel group of every plane lies inside the real extent, or it fetches nothing (the group index is
    // the same in every plane: 8 units per plane are a whole number of components)
    const bool x_grp_ok = !PAD || c0 + (s_src % UC) * 8 < g.Cq, d_grp_ok = !PAD || f0 + (s_src % UF) * 8 < g.F;
    const unsigned xs2 = (unsigned)((int)g.x_ss[2]) * 2u, dys = (unsigned)((int)g.dy_ss) * 2u;               // position -> position
    // ---- wave-uniform line state: a K step's rows lie on padded line A (the line of its row 0) or on B = A + 1
    int su0, Pj, lineB, nB, o0B, o1B;
    unsigned xbA, xbB, dbA, dbB;
    int xvA, xvB, dvA, dvB;
    auto line_vals = [&](int line, int n, int o0, int o1, unsigned &xb, int &xv, unsigned &db, int &dv) {
        const int i0 = o0 * g.pa[0] + t0 * g.pb[0] + g.pc[0], i1 = o1 * g.pa[1] + t1 * g.pb[1] + g.pc[1];
        dv = line < g.b_nlines;
        xv = dv && (unsigned)i0 < (unsigned)g.isp[0] && (unsigned)i1 < (unsigned)g.isp[1];
        xb = (unsigned)(n * (int)g.x_sn + i0 * (int)g.x_ss[0] + i1 * (int)g.x_ss[1] + g.b_cshift * (int)g.x_ss[2]) * 2u;
        db = (unsigned)(line * W * (int)g.dy_ss) * 2u;
    };
    {
        const int line = p_begin / WP;
        su0 = p_begin - line * WP; Pj = p_begin;
        int l = line;
        const int o1 = l % g.osp[1]; l /= g.osp[1];
        const int o0 = l % g.osp[0];
        const int n = l / g.osp[0];
        line_vals(line, n, o0, o1, xbA, xvA, dbA, dvA);
        lineB = line; nB = n; o0B = o0; o1B = o1;
    }
    auto next_line = [&]() {
        ++lineB; ++o1B;
        if (o1B == g.osp[1]) { o1B = 0; ++o0B; if (o0B == g.osp[0]) { o0B = 0; ++nB; } }
        line_vals(lineB, nB, o0B, o1B, xbB, xvB, dbB, dvB);
    };
    next_line();
    auto advance_tile = [&]() {
        su0 += KM; Pj += KM;
        if (su0 >= WP) { su0 -= WP; xbA = xbB; xvA = xvB; dbA = dbB; dvA = dvB; next_line(); }
    };
    // byte offsets of the units this thread fetches for the tile the line state points at: its X row, its dY row, and (wave 0)
    // its halo row KM + s_row
    unsigned vx = kOOR, vd = kOOR, vh = kOOR;
    auto decode_x = [&](int r) -> unsigned {
        const int t = su0 + r;
        const bool selB = t >= WP;
        const int u = selB ? t - WP : t;
        const bool ok = (selB ? xvB : xvA) != 0 && (unsigned)(u + g.b_cshift) < (unsigned)g.isp[2] && Pj + r < p_end + KIN - 1 && x_grp_ok;
        return ok ? (selB ? xbB : xbA) + (unsigned)u * xs2 + x_thr : kOOR;
    };
    auto decode_rows = [&]() {
        vx = decode_x(s_row);
        {
            const int t = su0 + s_row;
            const bool selB = t >= WP;
            const int u = selB ? t - WP : t;
            const bool ok = (selB ? dvB : dvA) != 0 && u < W && Pj + s_row < p_end && d_grp_ok;
            vd = ok ? (selB ? dbB : dbA) + (unsigned)u * dys + d_thr : kOOR;
        }
        if (wave == 0) vh = decode_x(KM + s_row);
    };
    const int dma_w = wave * 1024;                  // this wave's 8 rows of a plane
#define QK_DMA_X(BUF)  do { _Pragma("unroll") for (int p_ = 0; p_ < NPX; ++p_) \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t *)((lds_char *)(BUF) + p_ * XPL + dma_w), 16, (int)vx, (int)(p_ * x_pstep), 0, 0); } while (0)
#define QK_DMA_H(BUF)  do { if (wave == 0) { _Pragma("unroll") for (int p_ = 0; p_ < NPX; ++p_) \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void_t *)((lds_char *)(BUF) + p_ * XPL + KM * 128), 16, (int)vh, (int)(p_ * x_pstep), 0, 0); } } while (0)
#define QK_DMA_D1(BUF, P_) __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lds_void_t *)((lds_char *)(BUF) + DOFF + (P_) * DPL + dma_w), 16, (int)vd, (int)((P_) * d_pstep), 0, 0)

    floatx16 acc[KIN][CTW];
#pragma unroll
    for (int t = 0; t < KIN; ++t)
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][ct][r] = 0.f;
    float dbacc[CTW];
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) dbacc[ct] = 0.f;

    // ---- fragment addresses: lane -> (row fr_row of the 16 a transposing read pair covers, 8-byte chunk fr_ch) --------------
    const int Lg = lane & 15, g16 = (lane >> 4) & 1, kh = lane >> 5;
    const int fr_row = 8 * kh + (Lg >> 2);
    const int fr_ch = 16 * g16 + 4 * (Lg & 3);
    // band row = c + fr_row with c = 16 ks + t known at compile time; the row's block swap is ((c + fr_row) >> 1) & 1
    //   = ((c >> 1) & 1) ^ (((fr_row + (c & 1)) >> 1) & 1): four per-lane bases (c & 1, (c >> 1) & 1), everything else an immediate
    int a_base[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int e = v >> 1, mb = v & 1;
        const int sw = (((fr_row + e) >> 1) & 1) ^ mb;
        a_base[v] = (rt >> 1) * XPL + fr_row * 128 + (((rt & 1) ^ sw) * 64) + fr_ch * 2;
    }
    int b_base[CTW];
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) {               // column tile cg CTW + ct = 64 bytes of the dY row: plane (tile >> 1), block (tile & 1)
        const int ti = cg * CTW + ct;
        b_base[ct] = DOFF + (ti >> 1) * DPL + fr_row * 128 + (((ti & 1) ^ ((fr_row >> 1) & 1)) * 64) + fr_ch * 2;
    }

#define QK_FRAG_A(FA, BUF, KS_, T_) \
        FA[T_] = tr_pair((const lds_char *)(BUF) + a_base[((T_) & 1) * 2 + (((T_) >> 1) & 1)] + ((KS_) * 16 + (T_)) * 128)
#define QK_FRAG_B(FB, BUF, KS_) do { \
        _Pragma("unroll") for (int c_ = 0; c_ < CTW; ++c_) FB[c_] = tr_pair((const lds_char *)(BUF) + b_base[c_] + (KS_) * 16 * 128); \
    } while (0)

    // A K step = two halves: H0 = sub-steps 0, 1, then the wait for tile it + 1 and the barrier; H1 = sub-steps 2, 3, which ask for
    // tile it + 2.  The loop body is [H1(it) H0(it + 1)], three per trip: its back edge sits right behind a vmcnt(0), so hipcc's
    // waitcnt pass sees NO LDS-DMA pending at the loop header (with the back edge between two K steps it merged "pending into
    // some buffer" from the back edge with the prologue's state and put a vmcnt(0) in front of the first fragment read of the
    // trip: the tile then had half a K step to arrive, not a whole one); for the same reason there is no early exit inside a
    // trip -- the number of K steps is padded to 3 n + 1 with tiles past p_end, whose rows are out-of-range loads (zeros).
    const int iters = (p_end - p_begin + KM - 1) / KM;
    const int steps = iters <= 0 ? 0 : (iters + 1) / 3 * 3 + 1;          // smallest 3 n + 1 >= iters
    // Fragments: the dY fragments of sub-step s + 1 are read at the head of sub-step s into the other of two sets; the X fragment
    // of tap t is re-read for sub-step s + 1 right behind its last MFMA of sub-step s, into the registers it has just left (a
    // second set of all taps' fragments would be 20 more registers: 9 spilled) -- eight MFMAs of cover for every read.
    v8s fA[KIN], fB[2][CTW];
    bool mine = false;
#define QK_SUBSTEP(KS_, BCUR, BNXT, BDMA, LAST) do { \
        constexpr int cur_ = (KS_) & 1, nx_ = cur_ ^ 1;                       /* (KS is even: the sets alternate across steps too) */ \
        if ((KS_) + 1 < KS) QK_FRAG_B(fB[nx_], BCUR, (KS_) + 1); \
        else if (!(LAST)) QK_FRAG_B(fB[nx_], BNXT, 0); \
        __builtin_amdgcn_sched_barrier(0); \
        if ((KS_) == 0) { mine = turn == 0 && bias_wave; turn = turn == 0 ? n_share - 1 : turn - 1; decode_rows(); }   /* rows of tile it + 2 */ \
        if ((KS_) == 1) advance_tile(); \
        if (mine) { \
            _Pragma("unroll") for (int ct = 0; ct < CTW; ++ct) { \
                typedef unsigned u4v __attribute__((ext_vector_type(4))); \
                const u4v w = __builtin_bit_cast(u4v, fB[cur_][ct]); \
                dbacc[ct] = dot_ones(T(), w.x, dbacc[ct]); dbacc[ct] = dot_ones(T(), w.y, dbacc[ct]); \
                dbacc[ct] = dot_ones(T(), w.z, dbacc[ct]); dbacc[ct] = dot_ones(T(), w.w, dbacc[ct]); \
            } \
        } \
        _Pragma("unroll") for (int j = 0; j < NMF; ++j) { \
            const int t = j / CTW, ct = j % CTW; \
            acc[t][ct] = mfma16b(T(), fA[t], fB[cur_][ct], acc[t][ct]); \
            if (ct == CTW - 1) { \
                if ((KS_) + 1 < KS) QK_FRAG_A(fA, BCUR, (KS_) + 1, t); \
                else if (!(LAST)) QK_FRAG_A(fA, BNXT, 0, t); \
            } \
            if (!(LAST)) { \
                if ((KS_) == 2) { if (j == (NMF > 1 ? 1 : 0)) QK_DMA_X(BDMA); if (j == (NMF > 3 ? 3 : NMF - 1)) QK_DMA_H(BDMA); } \
                if ((KS_) == 3) {                                     /* the NPD dY planes spread over the sub-step's NMF slots */ \
                    _Pragma("unroll") for (int p_ = 0; p_ < NPD; ++p_) if ((p_ * NMF + NMF / 2) / NPD == j) QK_DMA_D1(BDMA, p_); \
                } \
            } \
            __builtin_amdgcn_sched_barrier(0); \
        } \
        if ((KS_) == 1) { \
            /* tile it + 1 (asked for one K step ago) has landed for this wave; behind the barrier for everybody, and everybody \
               has left tile it - 1: its buffer takes tile it + 2 from sub-step 2 on */ \
            __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));          /* vmcnt(0) */ \
            __builtin_amdgcn_s_barrier(); \
        } \
    } while (0)
    if (steps > 0 && !(g.ablate & 4)) {
        // ---- prologue: tiles 0 and 1 on their way, tile 0 landed behind the barrier, its first fragments read
        decode_rows(); advance_tile();
        QK_DMA_X(lds0); QK_DMA_H(lds0);
#pragma unroll
        for (int p = 0; p < NPD; ++p) QK_DMA_D1(lds0, p);
        decode_rows(); advance_tile();
        QK_DMA_X(lds1); QK_DMA_H(lds1);
#pragma unroll
        for (int p = 0; p < NPD; ++p) QK_DMA_D1(lds1, p);
        __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));          // vmcnt(0) (tile 1 rides along: once per block)
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int t = 0; t < KIN; ++t) QK_FRAG_A(fA, lds0, 0, t);
        QK_FRAG_B(fB[0], lds0, 0);
        QK_SUBSTEP(0, lds0, lds1, lds2, false); QK_SUBSTEP(1, lds0, lds1, lds2, false);                  // H0(0)
        for (int it = 0; it + 1 < steps; it += 3) {
            QK_SUBSTEP(2, lds0, lds1, lds2, false); QK_SUBSTEP(3, lds0, lds1, lds2, false);              // H1(it)
            QK_SUBSTEP(0, lds1, lds2, lds0, false); QK_SUBSTEP(1, lds1, lds2, lds0, false);              // H0(it + 1)
            QK_SUBSTEP(2, lds1, lds2, lds0, false); QK_SUBSTEP(3, lds1, lds2, lds0, false);
            QK_SUBSTEP(0, lds2, lds0, lds1, false); QK_SUBSTEP(1, lds2, lds0, lds1, false);
            QK_SUBSTEP(2, lds2, lds0, lds1, false); QK_SUBSTEP(3, lds2, lds0, lds1, false);
            QK_SUBSTEP(0, lds0, lds1, lds2, false); QK_SUBSTEP(1, lds0, lds1, lds2, false);              // H0(it + 3)
        }
        QK_SUBSTEP(2, lds0, lds1, lds2, true); QK_SUBSTEP(3, lds0, lds1, lds2, true);                    // H1(steps - 1)
        __builtin_amdgcn_s_waitcnt(0);
    }
#undef QK_SUBSTEP
    __syncthreads();
#undef QK_FRAG_A
#undef QK_FRAG_B
#undef QK_DMA_X
#undef QK_DMA_H
#undef QK_DMA_D1

    // ---- fold the 16 expanded blocks onto the 4 compact parts (as k_wgrad16_band; the slab of tap t lives in buffer t / 2) ----
    if ((g.ablate & 1) && acc[0][0][0] != 123.456f) return;
    const int lr = lane & 31;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {                  // phase = gathered component a
        const bool active = RTT == 2 ? rt == (ph >> 1) : rt == ph;
        if (active) {
#pragma unroll
            for (int ct = 0; ct < CTW; ++ct) {
                const int col = (cg * CTW + ct) * 32 + lr;
                const int b = col / BF, ff = col % BF;
                const int p = ph ^ b;
                const bool neg = (g.sign_tbl >> (ph * 4 + b)) & 1u;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (RTT == 2 && (r >> 3) != (ph & 1)) continue;      // this register's rows belong to the other component
                    const int row = mfma32_row(r, lane);
                    const int cc = RTT == 2 ? (row & 15) : row;
#pragma unroll
                    for (int t = 0; t < KIN; ++t) {
                        float *slab = reinterpret_cast<float *>(t / 2 == 0 ? lds0 : t / 2 == 1 ? lds1 : lds2) + (t & 1) * (TAPB / 4);
                        const float v = neg ? -acc[t][ct][r] : acc[t][ct][r];
                        float *dst = &slab[(cc * 4 + p) * BF + ff];
                        *dst = ph == 0 ? v : *dst + v;
                    }
                }
            }
        }
        __syncthreads();
    }
    const int tap0 = ot * KIN;                        // compact tap index of inner tap 0
#pragma unroll
    for (int t = 0; t < KIN; ++t) {
        const float *slab = reinterpret_cast<const float *>(t / 2 == 0 ? lds0 : t / 2 == 1 ? lds1 : lds2) + (t & 1) * (TAPB / 4);
        for (int e = tid; e < CQB * 4 * BF; e += NTHR) {
            const int ff = e % BF;
            const int p = (e / BF) & 3;
            const int cc = e / (4 * BF);
            if (!(g.ablate & 2) && (!PAD || (c0 + cc < g.Cq && f0 + ff < g.F)))
                atomicAdd(dw + (((tap0 + t) * g.Cq + c0 + cc) * 4 + p) * g.F + f0 + ff, slab[e]);
        }
    }
    if (bias_wave) {
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct) {
            const float s = dbacc[ct] + __shfl_xor(dbacc[ct], 32);            // the two 8-position halves of a K slice
            const int col = (cg * CTW + ct) * 32 + lr;
            if (lane < 32 && (!PAD || f0 + col % BF < g.F)) atomicAdd(dbias + (col / BF) * g.F + f0 + col % BF, s);
        }
    }
}

template <typename T, int RTT, int KIN, int CTW = 2, bool LINEAR_NARROW = false>
int run_wgrad16_band(const T *x, const T *dy, const T *ymask, float *dw, float *dbias, WgradGeom g, hipStream_t stream)
{
    constexpr int CQB = RTT * 8, BF = (8 / RTT) * CTW * 8, KM = 64;
    const bool padded = g.Cq % CQB != 0 || g.F % BF != 0;      // multiples of 16 only: the PAD form of the linear kernel (go_wgrad16_band admits nothing else)
    const long long other = (long long)g.ks[0] * g.ks[1] * ((g.Cq + CQB - 1) / CQB) * ((g.F + BF - 1) / BF);
    const long long total_p = (long long)g.b_nlines * g.b_wp;
    const long long max_splits = (total_p + KM - 1) / KM;
    const int slots = device_cu_count();              // 84 - 100 KB of LDS: one workgroup per CU
    const long long kEpilogueSteps = 24;              // fold + atomics, in K-step equivalents
    long long splits = 1, best_cost = -1;
    for (int r = 1; r <= 4; ++r) {
        long long sp = (long long)r * slots / other;
        sp = sp < 1 ? 1 : (sp > max_splits ? max_splits : sp);
        long long mps_r = (total_p + sp - 1) / sp;
        mps_r = (mps_r + KM - 1) / KM * KM;
        sp = (total_p + mps_r - 1) / mps_r;
        const long long rounds = (sp * other + slots - 1) / slots;
        const long long cost = rounds * (mps_r / KM + kEpilogueSteps);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; splits = sp; }
    }
    g.deterministic = (debug_flags() & kDbgDeterministic) ? 1 : 0;
    if (g.deterministic) splits = 1;                  // one block per gradient tile: one addition per element of dw
    long long mps = (total_p + splits - 1) / splits;
    mps = (mps + KM - 1) / KM * KM;
    splits = (total_p + mps - 1) / mps;
    g.m_per_split = (int)mps;
    g.n_splits = (int)splits;
    g.x_bytes = (unsigned)((long long)g.batch * g.x_sn * 2);
    g.dy_bytes = (unsigned)((long long)g.M * g.dy_ss * 2);
    g.ablate = debug_ablate();
    const long long n_tiles = splits * other;
    dim3 grid((unsigned)((n_tiles + 7) / 8 * 8), 1, 1);
    // (RTT 2 / KIN 5: the masked form exists only with one column tile per wave, the linear form only with two -- go_wgrad16_band)
    constexpr bool kMasked = !(RTT == 2 && KIN == 5 && CTW == 2) && !LINEAR_NARROW, kLinear = !(RTT == 2 && KIN == 5 && CTW == 1) && !LINEAR_NARROW;
    constexpr bool kNarrow = RTT == 2 && CTW == 1 && LINEAR_NARROW;            // the round-5 16 x 32 block form (linear kernel only)
    if (padded || kNarrow) {
        if constexpr ((RTT == 4 && CTW == 2) || kNarrow) {
            if (g.has_mask || g.dym || g.b_wp < KM + 8) return 0;          // (the caller goes on: fp32-MFMA kernels)
            if (padded) hipLaunchKernelGGL((k_wgrad16_band3<T, RTT, KIN, true, CTW>), grid, dim3(512), 0, stream, x, dy, dw, dbias, g);
            else hipLaunchKernelGGL((k_wgrad16_band3<T, RTT, KIN, false, CTW>), grid, dim3(512), 0, stream, x, dy, dw, dbias, g);
        } else return 0;
    } else if (g.has_mask) {
        if constexpr (kMasked) hipLaunchKernelGGL((k_wgrad16_band<T, RTT, KIN, true, CTW>), grid, dim3(512), 0, stream, x, dy, ymask, dw, dbias, g);
        else return QK_ERR_UNSUPPORTED;
    } else if (CTW == 2 && !g.dym && g.b_wp >= KM + 8 && !(debug_flags() & kDbgWgradBandV1)) {
        // the linear form: LDS-DMA staging, three tile buffers (round 5; QK_DBG_WGRAD_BAND_V1 / env QK_WGRAD_BAND_V1 = the round-2..4 kernel, A/B)
        if constexpr (CTW == 2) hipLaunchKernelGGL((k_wgrad16_band3<T, RTT, KIN>), grid, dim3(512), 0, stream, x, dy, dw, dbias, g);
    } else {
        if constexpr (kLinear) hipLaunchKernelGGL((k_wgrad16_band<T, RTT, KIN, false, CTW>), grid, dim3(512), 0, stream, x, dy, ymask, dw, dbias, g);
        else return QK_ERR_UNSUPPORTED;
    }
    return hipGetLastError() == hipSuccess ? 1 : QK_ERR_LAUNCH;
}

// Band geometry of a backward-weight call, or false: the innermost used axis must have 3 or 5 taps with unit stride
// and dilation; axes are rotated so that it is index 2 (unit axes move to the front: neither the position order nor
// the tap order changes).
bool wgrad_band_geom(const WgradGeom &g, WgradGeom *o)
{
    int ax = 2;
    while (ax > 0 && g.osp[ax] == 1 && g.isp[ax] == 1 && g.ks[ax] == 1) --ax;
    if (g.ks[ax] != 3 && g.ks[ax] != 5) return false;
    if (g.pa[ax] != 1 || g.pb[ax] != 1) return false;
    *o = g;
    const int sh = 2 - ax;
    for (int i = 0; i < 3; ++i) {
        const int src = i - sh;
        o->osp[i] = src >= 0 ? g.osp[src] : 1; o->isp[i] = src >= 0 ? g.isp[src] : 1; o->ks[i] = src >= 0 ? g.ks[src] : 1;
        o->pa[i] = src >= 0 ? g.pa[src] : 1; o->pb[i] = src >= 0 ? g.pb[src] : 1; o->pc[i] = src >= 0 ? g.pc[src] : 0;
        o->x_ss[i] = src >= 0 ? g.x_ss[src] : 0;
    }
    const int k = o->ks[2];
    o->b_wp = o->osp[2] + k - 1;
    if ((k - 1) * 12 > o->b_wp) return false;                // > 8 % of the positions would be padding
    const long long lines = (long long)g.batch * o->osp[0] * o->osp[1];
    if (lines * o->b_wp >= (1ll << 31) - 512) return false;
    o->b_nlines = (int)lines;
    o->b_cshift = o->pc[2];
    return true;
}

template <typename T>
int go_wgrad16_band(const void *x, const void *dy, const void *ymask, float *dw, float *dbias, const WgradGeom &g,
                    hipStream_t stream)
{
    const T *xp = (const T *)x, *dp = (const T *)dy, *yp = (const T *)ymask;
    const bool k5 = g.ks[2] == 5;
    // F % 64 == 0: 16 channels x 64 filters per block -- except with the relu mask, where the narrower dY tile of the
    // 32 x 32 form halves the mask work per block (B = 256 body layer: 1117 vs 1135 us linear, 1350 vs 1221 us masked)
    if (g.F % 64 == 0 && g.Cq % 16 == 0 && !(g.has_mask && g.Cq % 32 == 0)) {
        // masked, five taps, Cq a multiple of 16 only: the 64-filter block with its eight staged mask / dY units per thread
        // does not fit 256 registers (it spilled); ONE column tile per wave (16 channels x 32 filters per block, five
        // accumulator tiles) does -- a rare shape, never a chain's (their dY arrives pre-masked)
        if (g.has_mask && k5) return run_wgrad16_band<T, 2, 5, 1>(xp, dp, yp, dw, dbias, g, stream);
        return k5 ? run_wgrad16_band<T, 2, 5>(xp, dp, yp, dw, dbias, g, stream) : run_wgrad16_band<T, 2, 3>(xp, dp, yp, dw, dbias, g, stream);
    }
    // Cq = 16 (mod 32) with a filter count that is not a multiple of 64 (start_filter = 16 models: 16 -> 16, 16 -> 32), no mask:
    // 16 channels x 32 filters per block -- exact on the channel side, padded to 32 filters at most
    if (g.Cq % 32 != 0 && !g.has_mask)
        return k5 ? run_wgrad16_band<T, 2, 5, 1, true>(xp, dp, yp, dw, dbias, g, stream) : run_wgrad16_band<T, 2, 3, 1, true>(xp, dp, yp, dw, dbias, g, stream);
    return k5 ? run_wgrad16_band<T, 4, 5>(xp, dp, yp, dw, dbias, g, stream) : run_wgrad16_band<T, 4, 3>(xp, dp, yp, dw, dbias, g, stream);
}

}  // namespace

// Returns 1 when the band kernel took the call, 0 when the shape is outside it (the caller goes on to k_wgrad16),
// < 0 on error.  dw / dbias must already be zeroed (atomic accumulation).
int try_wgrad_band_16(int dtype, const void *x, const void *dy, const void *ymask, float *dw, float *dbias,
                      const WgradGeom &g, hipStream_t stream)
{
    if (dtype != QK_BF16 && dtype != QK_F16) return 0;
    if (debug_flags() & (kDbgNoMfma16 | kDbgNoWgradBand)) return 0;
    if (g.x_sc != 1 || g.dy_sc != 1) return 0;                         // channels_last buffers only
    const long long S = (long long)g.osp[0] * g.osp[1] * g.osp[2];
    if (g.dy_sn != S * g.dy_ss) return 0;                              // dy rows are addressed by flat position
    if (g.F % 16 != 0 || g.Cq % 16 != 0) return 0;                     // (multiples of 16 but not of 32: the PAD form of the 32 x 32 block kernel, round 5)
    if (g.has_mask && (g.F % 32 != 0 || g.Cq % (g.F % 64 == 0 ? 16 : 32) != 0)) return 0;
    if ((long long)g.batch * g.x_sn * 2 >= 0xF0000000ll || (long long)g.M * g.dy_ss * 2 >= 0xF0000000ll) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(ymask)) & 15) return 0;
    WgradGeom bg;
    if (!wgrad_band_geom(g, &bg)) return 0;
    if ((long long)bg.taps * bg.Cq * 4 * bg.F >= (1ll << 31)) return 0;
    note_path(QK_PATH_MFMA16_BAND);
    if (dtype == QK_BF16) return go_wgrad16_band<bf16>(x, dy, ymask, dw, dbias, bg, stream);
    return go_wgrad16_band<f16>(x, dy, ymask, dw, dbias, bg, stream);
}

}  // namespace qk
