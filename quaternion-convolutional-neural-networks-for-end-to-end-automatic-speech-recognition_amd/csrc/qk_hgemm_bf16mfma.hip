// Hamilton implicit GEMM on the 16-bit-input matrix cores (v_mfma_f32_32x32x16_{bf16,f16}), gfx950.
// Forward and backward-data of the quaternion convolution / dense layers for bfloat16 / float16
// activations, fp32 accumulation.  (Backward-weight for 16-bit data: qk_wgrad_bf16mfma.hip.)
//
// Hamilton structure in REGISTERS: per 16-deep MFMA step a wave loads the 4 gathered-component
// A fragments (r,i,j,k of the same rows/channels) and the 4 compact-part B fragments once and
// issues the 16 MFMAs of the 4x4 block table: acc[b] += A[a] * B[a ^ b] for the ten positive
// entries, accn[b] += ... for the six negative ones (y = acc - accn in the epilogue: no sign
// manipulation in the loop at all).  Every fragment feeds 4 MFMAs; the
// 4x-expanded weight (conv.py:327-331) exists nowhere -- HBM and LDS hold the compact kernel only.
//
// Workgroup = 8 waves (two per SIMD, so one wave's ds_read latency hides under the other's
// MFMAs), WM x WN waves of 32 rows x (4 components x 32 channels).  K step = one tap x 32 gathered
// channels x 4 components:
//   A tile  BM rows x 256 B, row-major, 16-byte slots XOR-swizzled with (row & 15): staging writes
//           (8 consecutive lanes = 8 slots of a row) and fragment reads (16 lanes = 16 rows, one
//           slot) are both bank-conflict free;
//   B tile  [k-slot 4][part 4][BF channels] x 16 B, exactly as the prep kernel lays the compact
//           kernel out in the workspace, so staging is a linear copy and consecutive lanes read
//           consecutive 16-byte slots.
// Two LDS buffers + a register stage give a 3-deep pipeline with ONE barrier per K step: while
// buffer k feeds the MFMAs, tile k+1 moves registers -> LDS and tile k+2 is in flight from L2/HBM
// (issue-only loads with clamped addresses; zero fill and the relu mask are applied at LDS-store time).
#include "qk_common.h"
#include "qk_postop.h"
#include <type_traits>

namespace qk {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ floatx16 mfma16(bf16, const uint4 &a, const uint4 &b, const floatx16 &c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ floatx16 mfma16(f16, const uint4 &a, const uint4 &b, const floatx16 &c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// two floats -> one dword of two 16-bit values (lo = a, hi = b), round-to-nearest-even in hardware:
// v_cvt_pk_bf16_f32 on gfx950 (the bit-twiddling from_f32<bf16> costs ~6 VALU per element; the epilogue
// converts 64 elements per lane)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(bf16, float a, float b)
{
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ unsigned pack2(f16, float a, float b)
{
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}

// zero the 16-bit halves whose mask half is <= 0 (as a signed integer == as a bf16/fp16 value):
// v_pk_max_i16, v_pk_min_u16, v_pk_mul_lo_u16, v_and
typedef short s2v __attribute__((ext_vector_type(2)));
typedef unsigned short u2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned mask2(unsigned v, unsigned m)
{
    const s2v pos = __builtin_elementwise_max(__builtin_bit_cast(s2v, m), (s2v)(0));
    const u2v one = __builtin_elementwise_min(__builtin_bit_cast(u2v, pos), (u2v)(1));
    return v & __builtin_bit_cast(unsigned, (u2v)(one * (u2v)(0xffff)));
}
__device__ __forceinline__ uint4 mask8(const uint4 &v, const uint4 &m)
{
    return make_uint4(mask2(v.x, m.x), mask2(v.y, m.y), mask2(v.z, m.z), mask2(v.w, m.w));
}

// ---------------------------------------------------------------------------------------
// compact fp32 kernel -> 16-bit, laid out per (tap, 32-channel K chunk) as [slot][part][j][8]
//   forward  (transposed == 0): K index = input channel c, j = filter f
//   bwd-data (transposed == 1): K index = filter f,        j = input channel c
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
k_prep_w16(const float *__restrict__ w, T *__restrict__ wq, int taps, int Cq, int F, int transposed, int neg_ijk, int ch_major)
{
    // ch_major (qk_conv_desc_t.kernel_order): the compact kernel lies as (cq, taps, 4 fq) -- row (c, t) instead of (t, c)
    // (Q, J multiples of 16: the layout is written for the 32-channel granule of the kernels, zero beyond the real extents --
    //  round 5, channel counts that are multiples of 16 only)
    const int Qr = transposed ? F : Cq, Jr = transposed ? Cq : F;
    const int Q = (Qr + 31) / 32 * 32, J = (Jr + 31) / 32 * 32;
    const long long total = (long long)taps * Q * 4 * J;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        long long r = idx;
        const int e = r % 8; r /= 8;
        const int j = r % J; r /= J;
        const int p = r % 4; r /= 4;
        const int slot = r % 4; r /= 4;
        const int kc = r % (Q / 32);
        const int t = r / (Q / 32);
        const int k = kc * 32 + slot * 8 + e;
        const int c = transposed ? j : k;
        const int f = transposed ? k : j;
        const float v = (k < Qr && j < Jr) ? w[((long long)(ch_major ? c * taps + t : t * Cq + c) * 4 + p) * F + f] : 0.f;
        wq[idx] = from_f32<T>(neg_ijk && p ? -v : v);
    }
    if (blockIdx.x == 0 && threadIdx.x < 128) wq[total + threadIdx.x] = from_f32<T>(0.f);   // zero line for padding rows
}

// The same re-layout for up to 32 (layer, direction) jobs in ONE launch (qk_conv_prep_kernels): a training step re-lays
// every kernel out twice (forward and backward-data form) after each optimiser step -- 26 launches of ~5 us for the TIMIT
// model; batched, they are one.
template <typename T>
__global__ void __launch_bounds__(256)
k_prep_w16_batch(const PrepJobs jobs)
{
    const PrepJob &jb = jobs.j[blockIdx.y];
    const float *__restrict__ w = jb.w;
    T *__restrict__ wq = static_cast<T *>(jb.wq);
    const int Cq = jb.cq, F = jb.fq, transposed = jb.transposed, neg_ijk = jb.neg_ijk, ch_major = jb.ch_major, taps = jb.taps;
    if (jb.small) {                                   // the fragment layout of k_hconv16_small (qk_hconv16_small.hip)
        const int Qs = transposed ? F : Cq, Js = transposed ? Cq : F;
        const int q32 = Qs == 32, fb_n = Js / 16, kso = q32 ? jb.kin : (jb.kin + 1) / 2;
        const long long tot = (long long)jb.n_ot * kso * 4 * fb_n * 512;
        for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < tot; idx += (long long)gridDim.x * 256)
            prep_small16_write<T>(w, wq, idx, Cq, F, transposed, neg_ijk, jb.kin, q32, fb_n);
        if (blockIdx.x == 0 && threadIdx.x < 128) wq[tot + threadIdx.x] = from_f32<T>(0.f);
        return;
    }
    const int Qr = transposed ? F : Cq, Jr = transposed ? Cq : F;
    const int Q = (Qr + 31) / 32 * 32, J = (Jr + 31) / 32 * 32;                  // (zero-padded to the 32-channel granule, see k_prep_w16)
    const long long total = (long long)taps * Q * 4 * J;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        long long r = idx;
        const int e = r % 8; r /= 8;
        const int j = r % J; r /= J;
        const int p = r % 4; r /= 4;
        const int slot = r % 4; r /= 4;
        const int kc = r % (Q / 32);
        const int t = r / (Q / 32);
        const int k = kc * 32 + slot * 8 + e;
        const int c = transposed ? j : k;
        const int f = transposed ? k : j;
        const float v = (k < Qr && j < Jr) ? w[((long long)(ch_major ? c * taps + t : t * Cq + c) * 4 + p) * F + f] : 0.f;
        wq[idx] = from_f32<T>(neg_ijk && p ? -v : v);
    }
    if (blockIdx.x == 0 && threadIdx.x < 128) wq[total + threadIdx.x] = from_f32<T>(0.f);
}

// Workgroup-wide OR of a per-thread bit mask (bits [0, nbits)) over NW waves: one ballot per bit inside each
// wave, one LDS word per wave, one barrier.  (512 atomicOr on one LDS word serialise: ~7 us per tile, measured.)
template <int NW>
__device__ __forceinline__ unsigned wg_or_mask(unsigned mine, int nbits, unsigned *slots, int wave, int lane)
{
    static_assert(NW == 4 || NW == 8, "one 16-byte read per four waves");
    unsigned wmask = 0;
    for (int b = 0; b < nbits; ++b)
        if (__builtin_amdgcn_ballot_w64(((mine >> b) & 1u) != 0)) wmask |= 1u << b;
    if (lane == 0) slots[wave] = wmask;
    __syncthreads();
    const uint4 lo = *reinterpret_cast<const uint4 *>(slots);
    unsigned all = lo.x | lo.y | lo.z | lo.w;
    if constexpr (NW == 8) {
        const uint4 hi = *reinterpret_cast<const uint4 *>(slots + 4);
        all |= hi.x | hi.y | hi.z | hi.w;
    }
    return __builtin_amdgcn_readfirstlane(all);
}

// ---------------------------------------------------------------------------------------
template <typename T, int MT, int WM, int WN, bool CONJ, bool MASK>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))   // 256-register budget
k_hgemm16(const T *__restrict__ in, const T *__restrict__ mask, const uint4 *__restrict__ wq,
          const T *__restrict__ zero_line, const float *__restrict__ bias, T *__restrict__ out, const GemmGeom g)
{
    static_assert(WM * WN == 8, "8 waves per workgroup");
    constexpr int BM = WM * MT * 32;               // MT row tiles of 32 per wave
    constexpr int BF = WN * 32;
    constexpr bool SPLIT = MT == 1;                 // second accumulator set for the negative entries
    constexpr int BU = (16 * BF) / 512;             // 16-byte units of the B tile per thread
    static_assert(BM % 64 == 0 && (16 * BF) % 512 == 0, "tile/threads");
    constexpr unsigned TBL = CONJ ? kSignConj : kSignConv;
    constexpr int TILE_U = BM * 16 + 16 * BF;       // 16-byte units of one (A, B) tile pair
    __shared__ __attribute__((aligned(16))) uint4 lds[2 * TILE_U + 2];   // double buffered (+ 8 words: per-wave tap masks)
    // A: [row][slot ^ (row & 15)] at lds + buf*TILE_U ; B: [slot][part][j] right behind it

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch rule, used for speed
    // only); give every XCD a CONTIGUOUS range of row tiles so the halo rows neighbouring tiles
    // share (taps reach +-1 image row) are served by that XCD's own L2.
    const int n_mt = (g.M + BM - 1) / BM;
    const int per_xcd = (n_mt + 7) / 8;
    const int mtile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (mtile >= n_mt) return;
    const int m0 = mtile * BM;
    const int j0 = blockIdx.y * BF;
    const int nkc = g.Q / 32;

    // ---- staging ---------------------------------------------------------------------------
    // 8 threads per row, each moving 2 of the row's 16 slots (slot = s8 and s8 + 8): the 8 lanes a
    // ds_write_b128 is serviced in then hit 8 different 16-byte slots of one 128-byte bank window.
    // Address generation is hoisted out of the K loop: per row one base offset (tap 0 position,
    // possibly outside the tensor) and a bit mask of the taps that fall inside it; per K step the
    // wave-uniform tap displacement is added.  Padding rows read a zeroed 64-byte line instead of
    // being patched afterwards.
    constexpr int RPT2 = BM / 64;                   // rows per thread (passes of 64 rows)
    const int s_row = tid >> 3, s8 = tid & 7;
    // GEMM row -> (sample, output position); see GemmGeom::o0_major for the second row order
    struct RowPos { int n, o0, o1, o2; };
    auto row_of = [&g](int m) -> RowPos {
        RowPos r;
        const int q2 = fastdiv(m, g.dv_mul[0], g.dv_shr[0]);
        r.o2 = m - q2 * g.osp[2];
        const int q1 = fastdiv(q2, g.dv_mul[1], g.dv_shr[1]);
        r.o1 = q2 - q1 * g.osp[1];
        const int hi = fastdiv(q1, g.dv_mul[2], g.dv_shr[2]);
        r.n = g.o0_major ? q1 - hi * g.o0_major : hi;
        r.o0 = g.o0_major ? hi : q1 - hi * g.osp[0];
        return r;
    };
    int base_off[RPT2];
    unsigned tapmask[RPT2];
#pragma unroll
    for (int r = 0; r < RPT2; ++r) {
        int m = m0 + s_row + r * 64;
        base_off[r] = 0; tapmask[r] = 0;
        if (m < g.M) {
            const RowPos rp = row_of(m);
            const int n = rp.n, o0 = rp.o0, o1 = rp.o1, o2 = rp.o2;
            const int p0 = o0 * g.pa[0] + g.pc[0], p1 = o1 * g.pa[1] + g.pc[1], p2 = o2 * g.pa[2] + g.pc[2];
            base_off[r] = n * (int)g.in_sn + p0 * (int)g.in_ss[0] + p1 * (int)g.in_ss[1] + p2 * (int)g.in_ss[2];
            int t = 0;
            for (int t0 = 0; t0 < g.ks[0]; ++t0)
                for (int t1 = 0; t1 < g.ks[1]; ++t1)
                    for (int t2 = 0; t2 < g.ks[2]; ++t2, ++t) {
                        const int i0 = p0 + t0 * g.pb[0], i1 = p1 + t1 * g.pb[1], i2 = p2 + t2 * g.pb[2];
                        const bool ok = i0 >= 0 && i0 < g.isp[0] && i1 >= 0 && i1 < g.isp[1] && i2 >= 0 && i2 < g.isp[2];
                        tapmask[r] |= (ok ? 1u : 0u) << t;
                    }
        }
    }
    // Taps that no row of this tile can use contribute exact zeros: they are skipped altogether (K steps,
    // loads, MFMAs).  That is the whole backward-data of a 'valid' convolution whose kernel spans an axis
    // -- the TimeDistributed dense head run as an (F, 1) convolution: one of the F taps per row -- and the
    // border image rows of every 'same' convolution.  The mask is the OR of the rows' masks, via one LDS word.
    unsigned tile_taps;
    {
        unsigned mine = 0;
#pragma unroll
        for (int r = 0; r < RPT2; ++r) mine |= tapmask[r];
        tile_taps = wg_or_mask<8>(mine, g.taps, reinterpret_cast<unsigned *>(lds + 2 * TILE_U), wave, lane);
    }
    const int iters = __builtin_popcount(tile_taps) * nkc;
    // slot s8 / s8+8 -> (component, 8-channel group) -> element offset inside the row
    const int cmp_lo = s8 >> 2, cmp_hi = cmp_lo + 2, sub = (s8 & 3) * 8;
    static_assert(BU == 1 || BU == 2, "B prefetch registers are named, not an array: hipcc parks a\n"
                  "by-reference-captured array in LDS (promote-alloca) and waits for the load right away");
    uint4 ar[RPT2][2], mr[MASK ? RPT2 : 1][2], br0, br1;
    int lt0 = 0, lt1 = 0, lt2 = 0, lkc = 0, ltap = 0;      // (tap, K chunk) of the tile the loads fetch
    int ldelta = 0;                                         // its wave-uniform displacement
    const uint4 *lwsrc = wq;                                // and its slice of the re-laid-out kernel

    auto prep_loads = [&]() {
        ldelta = lt0 * g.pb[0] * (int)g.in_ss[0] + lt1 * g.pb[1] * (int)g.in_ss[1] +
                 lt2 * g.pb[2] * (int)g.in_ss[2] + lkc * 32 + sub;
        lwsrc = wq + (long long)(ltap * nkc + lkc) * 16 * g.J;
    };
    auto advance_loads = [&]() {                            // only called while a further valid tile exists
        if (++lkc == nkc) {
            lkc = 0;
            do {
                ++ltap;
                if (++lt2 == g.ks[2]) { lt2 = 0; if (++lt1 == g.ks[1]) { lt1 = 0; ++lt0; } }
            } while (!((tile_taps >> ltap) & 1u));
        }
    };
    if (tile_taps)                                          // first tap any row uses
        while (!((tile_taps >> ltap) & 1u)) {
            ++ltap;
            if (++lt2 == g.ks[2]) { lt2 = 0; if (++lt1 == g.ks[1]) { lt1 = 0; ++lt0; } }
        }
    auto load_a = [&](int r) {
        const bool ok = (tapmask[r] >> ltap) & 1u;
        const int off = base_off[r] + ldelta;
        const T *lo = ok ? in + off + cmp_lo * g.Q : zero_line;
        const T *hi = ok ? in + off + cmp_hi * g.Q : zero_line;
        ar[r][0] = *reinterpret_cast<const uint4 *>(lo);
        ar[r][1] = *reinterpret_cast<const uint4 *>(hi);
        if constexpr (MASK) {
            const T *mlo = ok ? mask + off + cmp_lo * g.Q : zero_line;
            const T *mhi = ok ? mask + off + cmp_hi * g.Q : zero_line;
            mr[r][0] = *reinterpret_cast<const uint4 *>(mlo);
            mr[r][1] = *reinterpret_cast<const uint4 *>(mhi);
        }
    };
    // B tile: 16 (slot, part) segments of BF 16-byte units each, J units apart in the workspace
    auto load_b = [&]() {
        br0 = lwsrc[(tid / BF) * g.J + j0 + tid % BF];
        if constexpr (BU == 2) br1 = lwsrc[((tid + 512) / BF) * g.J + j0 + (tid + 512) % BF];
    };
    auto store_a = [&](int r, int buf) {
        uint4 *As = lds + buf * TILE_U;
        const int row = s_row + r * 64;
        uint4 v0 = ar[r][0], v1 = ar[r][1];
        if constexpr (MASK) { v0 = mask8(v0, mr[r][0]); v1 = mask8(v1, mr[r][1]); }
        As[row * 16 + (s8 ^ (row & 15))] = v0;
        As[row * 16 + ((s8 + 8) ^ (row & 15))] = v1;
    };
    auto store_b = [&](int buf) {
        uint4 *Bs = lds + buf * TILE_U + BM * 16;
        Bs[tid] = br0;
        if constexpr (BU == 2) Bs[tid + 512] = br1;
    };
    // The counters never move past the last tile: once it has been fetched, further fetches repeat
    // it (the K loop issues its loads unconditionally; see below).
    int next_tile = 0;                                      // tile the counters point at
    auto advance_if_more = [&]() {
        if (next_tile + 1 < iters) { advance_loads(); ++next_tile; }
    };
    auto load_tile = [&]() {
        prep_loads();
#pragma unroll
        for (int r = 0; r < RPT2; ++r) load_a(r);
        load_b();
        advance_if_more();
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int r = 0; r < RPT2; ++r) store_a(r, buf);
        store_b(buf);
    };
    // staging op `op` of a K step: store A rows r, refill them; ...; store B, refill it.  The loop
    // below issues one op after every second MFMA of the step, so the global loads, the mask VALU and
    // the LDS stores run in the shadow of the matrix pipe instead of in a phase of their own.
    constexpr int N_OPS = 2 * RPT2 + 2;
    auto stage_op = [&](int op, int buf) {
        if (op < 2 * RPT2) {
            if (op % 2 == 0) store_a(op / 2, buf); else load_a(op / 2);
        } else if (op == 2 * RPT2) store_b(buf);
        else load_b();
    };

    floatx16 acc[MT][4], accn[SPLIT ? 4 : 1];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[mt][b][r] = 0.f; if (SPLIT) accn[b][r] = 0.f; }

    const int frow = wm * MT * 32 + lr;             // A-tile row this lane reads (+ 32*mt)
    const int fsw = frow & 15;
    const int a_rd0 = frow * 16;
    const int b_rd0 = BM * 16 + wn * 32 + lr;

    // Pipeline (one barrier per K step): registers hold tile it+1 (loaded during step it-1);
    //   step it:  MFMAs on buffer it&1, and between them: registers -> LDS buffer (it+1)&1, then the
    //             freed registers <- global loads of tile it+2;  barrier.
    // Buffer (it+1)&1 was last read in step it-1, which every wave left through that step's barrier.
    // The last two steps re-fetch (and re-store) the final tile instead of branching around the ops:
    // nobody reads that buffer again.
    load_tile();
    store_tile(0);
    if (iters > 1) load_tile();
    __syncthreads();

    for (int it = (g.ablate & 4) ? iters : 0; it < iters; ++it) {          // (ablate 4: profiling, no K loop)
        const int nb = (it + 1) & 1;
        const uint4 *a_rd = lds + (it & 1) * TILE_U + a_rd0;
        const uint4 *b_rd = lds + (it & 1) * TILE_U + b_rd0;
        prep_loads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 A[MT][4], B[4];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int a = 0; a < 4; ++a) A[mt][a] = a_rd[mt * 32 * 16 + ((a * 4 + ks * 2 + lh) ^ fsw)];
#pragma unroll
            for (int p = 0; p < 4; ++p) B[p] = b_rd[((ks * 2 + lh) * 4 + p) * BF];
            uint4 Bn[SPLIT ? 1 : 4];
            if constexpr (!SPLIT) {
                // taller wave tile: the three negated parts (12 v_xor) are shared by MT row tiles
#pragma unroll
                for (int p = 1; p < 4; ++p)
                    Bn[p] = make_uint4(B[p].x ^ 0x80008000u, B[p].y ^ 0x80008000u, B[p].z ^ 0x80008000u, B[p].w ^ 0x80008000u);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    constexpr unsigned tbl = TBL;
                    const bool ng = (tbl >> (a * 4 + b)) & 1u;
                    if constexpr (SPLIT) {
                        // products that enter with a minus sign go to a second accumulator set: no
                        // sign-flip VALU in the loop, acc -= accn at the end
                        if (ng) accn[b] = mfma16(T(), A[0][a], B[a ^ b], accn[b]);
                        else acc[0][b] = mfma16(T(), A[0][a], B[a ^ b], acc[0][b]);
                    } else {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt][b] = mfma16(T(), A[mt][a], ng ? Bn[a ^ b] : B[a ^ b], acc[mt][b]);
                    }
                    const int f = ks * 16 + a * 4 + b;           // MFMA group index inside the K step
                    if (f % 2 == 1 && f / 2 < N_OPS) stage_op(f / 2, nb);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        advance_if_more();
        __syncthreads();
    }

    // ---- epilogue: bias + activation, then 16-byte stores ------------------------------------
    // An MFMA accumulator holds ONE column per lane, so storing it directly means 64 two-byte stores
    // per lane (store-issue bound: 17 % of the kernel, measured).  Each wave instead transposes its
    // 32x32 tiles through a private 2.5 KB LDS patch (the tile buffers are free after the last
    // barrier; a wave's own LDS accesses execute in order) and writes 8 channels = 16 bytes per lane.
    if ((g.ablate & 8) && acc[0][0][0] != 123.456f) return;          // (ablate 8: profiling, no epilogue)
    constexpr int EP_PITCH = 80;                                   // 64 B of data + 16 B pad per row
    char *ep = reinterpret_cast<char *>(lds) + wave * (32 * EP_PITCH);
    const int e_row = lane >> 2, e_chunk = lane & 3;
    float bia4[4];                                                 // up front: see k_hgemm16_band
#pragma unroll
    for (int b = 0; b < 4; ++b) bia4[b] = g.has_bias ? bias[b * g.J + j0 + wn * 32 + lr] : 0.f;
    // post-op (PReLU / dropout, qk_postop.h): forward writes pre and y; backward-data applies the derivative
    const PostOp psd = resolve_seed(g.post);
    const bool post_on = g.post.kind != 0;
    const bool post_bwd = post_on && g.ep_mask != nullptr, post_fwd = post_on && g.post_fwd != 0;
    float *aslab = reinterpret_cast<float *>(reinterpret_cast<char *>(lds) + 32768);
    if (post_bwd && g.dalpha) {
        if (tid < 256) aslab[tid] = 0.f;
        __syncthreads();
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int a_key[2] = {0, 0};
        float a_val[2] = {0.f, 0.f}, dal[2] = {0.f, 0.f};
        if (post_on) {
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int m = m0 + (wm * MT + mt) * 32 + e_row + 16 * pass;
                if (m < g.M && g.post.alpha_sel >= 0) {
                    const RowPos rp = row_of(m);
                    const int o0 = rp.o0, o1 = rp.o1, o2 = rp.o2;
                    a_key[pass] = g.post.alpha_sel == 0 ? o0 : g.post.alpha_sel == 1 ? o1 : o2;
                }
                a_val[pass] = g.post.alpha ? g.post.alpha[a_key[pass]] : 0.f;
            }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int ch0 = b * g.J + j0 + wn * 32;
            const float bia = bia4[b];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {               // registers r, r + 1 hold consecutive rows
                float v0 = acc[mt][b][r] + bia, v1 = acc[mt][b][r + 1] + bia;
                if constexpr (SPLIT) { v0 -= accn[b][r]; v1 -= accn[b][r + 1]; }
                if (g.relu) { v0 = v0 > 0.f ? v0 : 0.f; v1 = v1 > 0.f ? v1 : 0.f; }
                const unsigned pk = pack2(T(), v0, v1);
                char *dst = ep + mfma32_row(r, lane) * EP_PITCH + lr * 2;
                *reinterpret_cast<unsigned short *>(dst) = (unsigned short)pk;
                *reinterpret_cast<unsigned short *>(dst + EP_PITCH) = (unsigned short)(pk >> 16);
            }
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int row = e_row + 16 * pass;
                const uint4 val = *reinterpret_cast<const uint4 *>(ep + row * EP_PITCH + e_chunk * 16);
                const int m = m0 + (wm * MT + mt) * 32 + row;
                if (m < g.M) {
                    int orow = m;
                    if (g.o0_major) {
                        const RowPos rp = row_of(m);
                        orow = ((rp.n * g.osp[0] + rp.o0) * g.osp[1] + rp.o1) * g.osp[2] + rp.o2;
                    }
                    const long long o = (long long)orow * (int)g.out_ss + ch0 + e_chunk * 8;
                    uint4 v = val;
                    if (g.ep_mask) {
                        const uint4 mk = *reinterpret_cast<const uint4 *>(static_cast<const T *>(g.ep_mask) + o);
                        if (post_bwd) v = post_bwd8<T>(v, mk, a_val[pass], (unsigned)o, psd, dal[pass]);
                        else v = mask8(v, mk);
                    }
                    if (post_fwd) {
                        if (g.pre_out) *reinterpret_cast<uint4 *>(static_cast<T *>(g.pre_out) + o) = v;
                        v = post_fwd8<T>(v, a_val[pass], (unsigned)o, psd);
                    }
                    *reinterpret_cast<uint4 *>(out + o) = v;
                }
            }
        }
        if (post_bwd && g.dalpha) {
            wave_add_by_key(dal[0], a_key[0], aslab, lane);
            wave_add_by_key(dal[1], a_key[1], aslab, lane);
        }
    }
    if (post_bwd && g.dalpha) {
        __syncthreads();
        if (tid < g.post.alpha_len && aslab[tid] != 0.f) atomicAdd(g.dalpha + tid, aslab[tid]);
    }
}

// ---------------------------------------------------------------------------------------
// Band variant: the taps along the innermost spatial axis reuse ONE staged A tile.
//
// In k_hgemm16 every (tap, channel chunk) K step stages its own 128-row A tile, although the KIN
// taps of one kernel row read the same input rows shifted by one position: A staging is ~15 % of
// that kernel (ablation), B staging ~11 %.  Here the rows of a tile run over PADDED lines of the
// innermost axis (out extent + KIN - 1 positions per line; the extra positions are computed and
// dropped, 2 % at 200 columns), which makes "input position of (row, inner tap t)" exactly
// "band row + t" -- no per-(row, tap) validity, out-of-line positions are zero rows of the band.
// A group = (outer tap, channel chunk) stages one band of BM + KIN - 1 rows and runs KIN sub-steps
// off it; the band of the next group is fetched during the first sub-steps and stored during the
// following ones (two band buffers), B tiles are double-buffered per sub-step as before.
// ---------------------------------------------------------------------------------------
// buffer-resource loads (see qk_wgrad_bf16mfma.hip): 32-bit per-lane byte offset + wave-uniform offset,
// no 64-bit address arithmetic, and offsets past the extent read as zeros (padding rows need no select
// between a data pointer and a zero line)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOutOfRange16 = 0xF0000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc16(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16b(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// 16-byte buffer store.  The wave-uniform part of the offset is ADDED to the per-lane offset instead of going into the
// instruction's soffset SGPR: hipcc (ROCm 7.2) inserts the wait states between a > 8-byte VMEM store and a VALU write
// of its data registers only when soffset is not a register (the documented exception), and on gfx950 a
// `buffer_store_dwordx4 v[a:a+3], ..., sN offen` directly followed by a VALU write of v[a] stored the NEW value in
// part of the lanes (lanes 12-15 of every 16: measured, k_hgemm16_point with an epilogue mask).
__device__ __forceinline__ void buf_store16b(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, const uint4 &v)
{
    const u32x4 d = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)(voff + soff), 0, 0);
}

// (outer_tap_mask: qk_common.h)
// A-band staging schedule of one group: op code q < R fetches row pass q of the NEXT band (an LDS-DMA into the other band buffer
// since the end of round 4; through registers before), -1 = nothing.  Codes R <= q < 2R were the register form's LDS stores --
// at least one sub-step after their loads -- and are no-ops now: the table still says WHEN a pass is asked for, which is what the
// vmcnt counts in front of the barriers are computed from.
#ifndef QK_BAND_PIPE
#define QK_BAND_PIPE 1
#endif
// probe builds (tools/probe/build_variant.sh probe -DQK_BAND_PROBE): QK_ABLATE 16 = no waits / barriers in the K loop, 32 = no DMA issued in the K loop
#ifdef QK_BAND_PROBE
#define QK_PROBE_ON(bit) (!(g.ablate & (bit)))
#else
#define QK_PROBE_ON(bit) true
#endif
constexpr int kBandOpsMax = 5;
constexpr int band_op(int R, int K, int ti, int k);
// row passes LOADED in sub-step ti of a group (op codes 0 .. R - 1), and whether the halo pass (R - 1) is among them: what the
// wait in front of a sub-step's barrier leaves in flight behind the B tile's LDS-DMA (k_hgemm16_band)
constexpr int band_loads_in(int R, int K, int ti) { int n = 0; for (int k = 0; k < kBandOpsMax; ++k) { const int q = band_op(R, K, ti, k); if (q >= 0 && q < R) ++n; } return n; }
constexpr bool band_loads_halo_in(int R, int K, int ti) { for (int k = 0; k < kBandOpsMax; ++k) if (band_op(R, K, ti, k) == R - 1) return true; return false; }
constexpr int band_op(int R, int K, int ti, int k)
{
    if (R == 3 && K == 5) { const int t[5][5] = {{0, 1, -1, -1, -1}, {3, 2, -1, -1, -1}, {4, -1, -1, -1, -1}, {5, -1, -1, -1, -1}, {-1, -1, -1, -1, -1}}; return t[ti][k]; }
    if (R == 3 && K == 3) { const int t[3][5] = {{0, 1, -1, -1, -1}, {3, 4, 2, -1, -1}, {5, -1, -1, -1, -1}}; return t[ti][k]; }
    if (R == 4 && K == 5) { const int t[5][5] = {{0, 1, -1, -1, -1}, {4, 2, -1, -1, -1}, {5, 3, -1, -1, -1}, {6, -1, -1, -1, -1}, {7, -1, -1, -1, -1}}; return t[ti][k]; }
    if (R == 4 && K == 3) { const int t[3][5] = {{0, 1, 2, -1, -1}, {4, 5, 3, -1, -1}, {6, 7, -1, -1, -1}}; return t[ti][k]; }
    if (R == 5 && K == 5) { const int t[5][5] = {{0, 1, -1, -1, -1}, {5, 2, -1, -1, -1}, {6, 3, -1, -1, -1}, {7, 4, -1, -1, -1}, {8, 9, -1, -1, -1}}; return t[ti][k]; }
    if (R == 5 && K == 3) { const int t[3][5] = {{0, 1, 2, -1, -1}, {5, 6, 7, 3, 4}, {8, 9, -1, -1, -1}}; return t[ti][k]; }
    return -2;
}

// Workgroup shapes (WM x WN waves of 32 rows x (4 components x 32 channels)):
//   8 waves, one workgroup per CU (LDS 100 - 150 KB): the original form;
//   4 waves, TWO workgroups per CU (LDS <= 80 KB each): a tile's prologue (row decode, first band: an HBM burst of
//     ~64 KB per CU) and epilogue (64 KB of stores) are 21 % (N = 256) to 38 % (N = 128) of the 8-wave kernels'
//     time (ablation), and with one workgroup per CU nothing overlaps them.  Two independent
//     workgroups per CU let the hardware run one's K loop under the other's prologue / epilogue.
//   TRIM (N = 128, 4 x 1 waves): the band buffer holds exactly BM = 128 rows, so a tile yields BM - (KIN - 1)
//     output rows and the last KIN - 1 rows of the wave tiles are computed and dropped (3 %): 2 x 128 rows x 256 B
//     + 2 B tiles = 80 KB is what lets two such workgroups share a CU.
//   PAD (round 5): channel counts that are multiples of 16 but not of 32 (start_filter = 16 models,
//     /root/reference/models/interspeech_model.py:46-50): the re-laid-out kernel is zero-padded to the 32-channel granule
//     (g.Qp, g.Jp), band units of channels >= Q are out-of-range DMA lanes (zeros), output pieces of channels >= J are not
//     stored.  The matrix cores then run 2x (one side padded) to 4x the algorithmic MFMAs -- still several times the fp32-MFMA
//     path these shapes took before; the unpadded instantiations are untouched.
template <typename T, int WM, int WN, int KIN, bool CONJ, bool TRIM, bool EPM, bool POSTF, bool PAD = false>
__global__ void __launch_bounds__(WM * WN * 64) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_hgemm16_band(const T *__restrict__ in, const uint4 *__restrict__ wq, const T *__restrict__ zero_line,
               const float *__restrict__ bias, T *__restrict__ out, const GemmGeom g)
{
    constexpr int NW = WM * WN, NTHR = NW * 64;
    static_assert(NW == 8 || NW == 4, "8 waves (one workgroup per CU) or 4 waves (two per CU)");
    static_assert(KIN >= 2 && KIN <= 9, "inner taps per band");
    constexpr int BM = WM * 32, BF = WN * 32;
    constexpr int BMU = TRIM ? BM - (KIN - 1) : BM;    // output rows a tile yields
    constexpr int BU = (16 * BF) / NTHR;               // 16-byte units of the B tile per thread
    static_assert(BU >= 1 && BU <= 4 && (16 * BF) % NTHR == 0 && NTHR % BF == 0, "B tile / threads");
    constexpr unsigned TBL = CONJ ? kSignConj : kSignConv;
    constexpr int BAND = TRIM ? BM : BM + KIN - 1;     // rows of the A band
    constexpr int A_U = (TRIM ? BM : BM + 8) * 16;     // 16-byte units of one band buffer
    constexpr int B_U = 16 * BF;
    // EVERYTHING the kernel stages goes L2 / HBM -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPR round trip, no ds_write; round 4,
    // +2.9 % in the isolated loop at unchanged clock and power; -0.13 ms on the step for the B tiles, -0.20 ms
    // more for the A band).  The two B buffers and the two band buffers are OBJECTS of their own and which ones a sub-step reads is a
    // compile-time constant (the group loop runs two groups per trip): hipcc orders every later ds_read of an LDS object behind a
    // pending LDS-DMA into it -- with one `lds[]` the fragment reads waited for vmcnt(0).
    // Band buffer: two PLANES [row][8 units of 16 bytes] (components 0, 1 | 2, 3): the 8 DMA lanes of a row fill its 128 bytes of a plane,
    // slot s of row r holding unit s ^ ((r >> 1) & 7) -- the swizzle sits on the SOURCE side (a DMA lane's slot is fixed) and keeps the
    // 16 rows of a fragment read on 16 different bank groups.  Rows outside the tensor: out-of-range offsets, the DMA writes zeros.
    __shared__ __attribute__((aligned(16))) uint4 ldsA0[A_U];
    __shared__ __attribute__((aligned(16))) uint4 ldsA1[A_U];
    constexpr int PL = A_U / 2;                        // one PLANE of a band buffer: [row][8 units] (components 0, 1 | 2, 3)
    __shared__ __attribute__((aligned(16))) uint4 ldsB0[B_U];
    __shared__ __attribute__((aligned(16))) uint4 ldsB1[B_U];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    const int WP = g.b_wp;
    const int total_p = g.b_nlines * WP;           // padded rows (host checked < 2^31)
    const int n_mt = (total_p + BMU - 1) / BMU;
    const int per_xcd = (n_mt + 7) / 8;
    const int mtile = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (mtile >= n_mt) return;
    const int p0 = mtile * BMU;
    const int j0 = blockIdx.y * BF;
    const int nkc = (PAD ? g.Qp : g.Q) / 32;
    const int JW = PAD ? g.Jp : g.J;                // filters per part of the re-laid-out kernel
#ifdef QK_PHASE_STAMPS       // probe builds only (tools/probe/phase_stamps.py): the stamps cost two registers, i.e. spills in the 64-row kernels
    unsigned long long *ts = (g.dbg_ts && blockIdx.x < 65536 && blockIdx.y == 0 && tid == 0) ? g.dbg_ts + 8 * blockIdx.x : nullptr;
    if (ts) ts[4] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);   // XCC_ID | HW_ID
#define QK_STAMP(i) do { if (ts) ts[i] = __builtin_readcyclecounter(); } while (0)
#else
#define QK_STAMP(i) do { } while (0)
#endif
    QK_STAMP(0);

    // ---- band rows of this thread (decoded once) ------------------------------------------------
    // staging: 8 threads (DMA lanes) per row, NTHR / 8 rows per pass
    constexpr int RPP = NTHR / 8;                   // rows per staging pass
    constexpr int RPT3 = (BAND + RPP - 1) / RPP;
    const int s_row = tid >> 3, s8 = tid & 7;
    int base_off[RPT3];
    unsigned omask[RPT3];                           // bit (t0 * ks1 + t1): outer tap inside the tensor
    // Outer taps no row of this band can use are skipped altogether (see k_hgemm16): border image rows.  Round 4: the set is
    // computed from the tile's GEOMETRY -- the padded lines it touches, wave-uniform scalar code -- instead of OR-ing the rows'
    // masks through LDS behind a barrier: the first band's loads no longer wait for every row to be decoded (they are issued
    // pass by pass, right behind each pass's decode, below).  A line whose positions inside the tile are all padding columns may
    // add a tap nobody uses: its loads read zeros, nothing else changes.
    unsigned tile_ot = 0;
    // Round 6: WAVE-UNIFORM LINE STATE (as k_wgrad16_band3 decodes its rows).  A band of BAND rows touches at most two padded lines
    // when BAND <= WP (every TIMIT layer: 68 / 128 rows against 204 positions per line): the two lines' sample / outer coordinates,
    // base offsets and outer-tap masks are decoded ONCE, in scalar registers; a thread's row then costs a compare, two selects and a
    // multiply-add instead of three divisions by multiply, the tap-range arithmetic and the offset polynomial (~ 55 VALU per row,
    // four rows per thread: 3.1 - 3.6 k of a 32-filter tile's 33 k cycles sat between the kernel's first instruction and its first
    // band load, profiles/r05_phase_stamps.txt).  Bands that span more lines (short inner axes) keep the per-row decode.
#ifndef QK_BAND_LINE_STATE
#define QK_BAND_LINE_STATE 1
#endif
    const bool two_lines = QK_BAND_LINE_STATE && BAND <= WP;
    int lb0 = 0, lb1 = 0, p_l0 = 0, p_l1 = 0x7fffffff;
    unsigned lm0 = 0, lm1 = 0;
    if (two_lines) {
        const int last_p = min(p0 + BAND - 1, total_p - 1);
        const int line_lo = fastdiv(p0, g.dv_mul[0], g.dv_shr[0]);
        auto line_state = [&](int line, int &lb, unsigned &lm) {
            const int l2 = fastdiv(line, g.dv_mul[1], g.dv_shr[1]), o1 = line - l2 * g.osp[1];
            const int n = fastdiv(l2, g.dv_mul[2], g.dv_shr[2]), o0 = l2 - n * g.osp[0];
            const int q0 = o0 * g.pa[0] + g.pc[0], q1 = o1 * g.pa[1] + g.pc[1];
            lb = __builtin_amdgcn_readfirstlane(n * (int)g.in_sn + q0 * (int)g.in_ss[0] + q1 * (int)g.in_ss[1]);
            lm = __builtin_amdgcn_readfirstlane(outer_tap_mask(q0, q1, g));
        };
        p_l0 = line_lo * WP;
        if (line_lo < g.b_nlines) line_state(line_lo, lb0, lm0);
        if (last_p >= p_l0 + WP && line_lo + 1 < g.b_nlines) { p_l1 = p_l0 + WP; line_state(line_lo + 1, lb1, lm1); }
        tile_ot = lm0 | lm1;
    } else {
        const int last_p = min(p0 + BAND - 1, total_p - 1);
        const int line_lo = fastdiv(p0, g.dv_mul[0], g.dv_shr[0]);
        const int line_hi = min(fastdiv(last_p, g.dv_mul[0], g.dv_shr[0]), g.b_nlines - 1);
        for (int line = line_lo; line <= line_hi; ++line) {
            const int l2 = fastdiv(line, g.dv_mul[1], g.dv_shr[1]), o1 = line - l2 * g.osp[1];
            const int n = fastdiv(l2, g.dv_mul[2], g.dv_shr[2]), o0 = l2 - n * g.osp[0];
            const int q0 = o0 * g.pa[0] + g.pc[0], q1 = o1 * g.pa[1] + g.pc[1];
            tile_ot |= outer_tap_mask(q0, q1, g);
        }
        tile_ot = __builtin_amdgcn_readfirstlane(tile_ot);
    }
    auto decode_row = [&](int r) {
        const int j = s_row + r * RPP;
        base_off[r] = 0; omask[r] = 0;
        const int P = p0 + j;
        if (two_lines) {
            // (rows past the tensor's last line: p_l1 stays out of reach or lm1 = 0 -- no tap, the DMA lane reads zeros)
            const bool in1 = P >= p_l1;
            const int col = P - (in1 ? p_l1 : p_l0) + g.b_cshift;
            if (j < BAND && col >= 0 && col < g.isp[2] && (in1 || P < p_l0 + WP)) {
                base_off[r] = (in1 ? lb1 : lb0) + col * (int)g.in_ss[2];
                omask[r] = in1 ? lm1 : lm0;
            }
            return;
        }
        const int line = fastdiv(P, g.dv_mul[0], g.dv_shr[0]);
        const int col = P - line * WP + g.b_cshift;
        if (j < BAND && line < g.b_nlines && col >= 0 && col < g.isp[2]) {
            const int l2 = fastdiv(line, g.dv_mul[1], g.dv_shr[1]), o1 = line - l2 * g.osp[1];
            const int n = fastdiv(l2, g.dv_mul[2], g.dv_shr[2]), o0 = l2 - n * g.osp[0];
            const int q0 = o0 * g.pa[0] + g.pc[0], q1 = o1 * g.pa[1] + g.pc[1];
            base_off[r] = n * (int)g.in_sn + q0 * (int)g.in_ss[0] + q1 * (int)g.in_ss[1] + col * (int)g.in_ss[2];
            omask[r] = outer_tap_mask(q0, q1, g);
        }
    };
    const int groups = __builtin_popcount(tile_ot) * nkc;
    const int substeps = groups * KIN;
    // LDS-DMA: lane (row, slot s8) fills slot s8 of its row; that slot HOLDS unit s8 ^ ((row >> 1) & 7) (the fragment reads' bank swizzle,
    // applied on the source side; 32 or 64 rows per pass: a constant of the thread)
    static_assert((NTHR / 8) % 16 == 0, "rows per pass");
    const int u_src = s8 ^ ((s_row >> 1) & 7);
    const int cmp_lo = u_src >> 2, sub = (u_src & 3) * 8;
    const __amdgpu_buffer_rsrc_t rin = make_rsrc16(in, g.b_in_bytes), rw = make_rsrc16(wq, g.b_w_bytes);
    const unsigned a_thr = (unsigned)(cmp_lo * g.Q + sub) * 2u;         // this thread's (component, 8 channels) of a row
    const unsigned a_hi = (unsigned)g.Q * 4u;                            // two components further (wave-uniform)
    // B unit u = tid + k * NTHR lies (NTHR / BF) (slot, part) segments further per k: a wave-uniform offset
    const unsigned b_thr0 = (unsigned)((tid / BF) * JW + j0 + tid % BF) * 16u;
    const unsigned b_kstep = (unsigned)((NTHR / BF) * JW) * 16u;
    constexpr bool HALO = !TRIM;                    // pass RPTF holds the KIN - 1 halo rows (none when trimmed)
    constexpr int RPTF = HALO ? RPT3 - 1 : RPT3;    // full passes
    static_assert(RPTF * RPP == BM && RPTF <= 4, "band = up to four full passes (+ one halo pass)");

    // group the A loads fetch (outer tap at0/at1 = index aot, channel chunk akc); stops at the last
    int at0 = 0, at1 = 0, aot = 0, akc = 0, a_next = 0, adelta = 0;
    auto a_prep = [&]() {
        adelta = at0 * g.pb[0] * (int)g.in_ss[0] + at1 * g.pb[1] * (int)g.in_ss[1] + akc * 32;
    };
    // next used outer tap = lowest set bit above the current one; (t0, t1) = divmod by ks[1] through a
    // 16-bit reciprocal (exact for these < 32 values).  Written loop-free on purpose: a data-dependent
    // loop here made hipcc keep base_off[] / omask[] in LDS and scratch.
    const unsigned inv_ks1 = (65536u + (unsigned)g.ks[1] - 1u) / (unsigned)g.ks[1];
    auto a_set_outer = [&](unsigned ot) {
        aot = (int)ot;
        at0 = (int)((ot * inv_ks1) >> 16);
        at1 = (int)ot - at0 * g.ks[1];
    };
    if (tile_ot) a_set_outer((unsigned)__builtin_ctz(tile_ot));         // first outer tap any row uses
    auto a_advance_if_more = [&]() {
        if (a_next + 1 < groups) {
            ++a_next;
            if (++akc == nkc) { akc = 0; a_set_outer((unsigned)__builtin_ctz(tile_ot & (~1u << aot))); }
        }
    };
    typedef __attribute__((address_space(3))) void lds_void;
    const int dma_slot = tid & ~63;
    // row pass r of the band -> band buffer object BUFOBJ: two DMAs (planes 0 and 1), lane = (row s_row + r RPP, slot s8)
#define QK_DMA_A(R, BUFOBJ) do { \
        const bool ok_ = ((omask[R] >> aot) & 1u) && (!PAD || akc * 32 + sub < g.Q); \
        const unsigned voff_ = ok_ ? (unsigned)(base_off[R] + adelta) * 2u + a_thr : kOutOfRange16; \
        if ((R) < RPTF || wave == 0) {               /* the halo pass holds KIN - 1 <= 8 rows: wave 0 only */ \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void *)((BUFOBJ) + (R) * RPP * 8 + dma_slot), 16, (int)voff_, 0, 0, 0); \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void *)((BUFOBJ) + PL + (R) * RPP * 8 + dma_slot), 16, (int)voff_, (int)a_hi, 0, 0); \
        } } while (0)
    // sub-step the B loads fetch: (outer tap bot, chunk bkc, inner step bti); stops at the last
    int bot = 0, bkc = 0, bti = 0, b_next = 0;
    unsigned bsoff = 0;
    auto b_prep = [&]() {
        const int tap = bot * KIN + bti;           // sub-step ti IS inner tap ti; b_rev only mirrors the band offset
        bsoff = (unsigned)((tap * nkc + bkc) * 16 * JW) * 16u;
    };
    if (tile_ot) bot = __builtin_ctz(tile_ot);
    auto b_advance_if_more = [&]() {
        if (b_next + 1 < substeps) {
            ++b_next;
            if (++bti == KIN) { bti = 0; if (++bkc == nkc) { bkc = 0; bot = __builtin_ctz(tile_ot & (~1u << bot)); } }
        }
    };
    // unit tid + k NTHR of the tile at bsoff -> its slot of a B buffer (lane-linear: wave base + lane x 16 bytes).  Sub-step s issues the DMA of
    // tile s + 1 FIRST (a whole sub-step to arrive) and waits for it -- vmcnt(the A loads issued behind it) -- in front of its barrier:
    // every wave's units are in LDS before anybody passes.
#define QK_DMA_B1(K, BUFOBJ) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void *)((BUFOBJ) + (K) * NTHR + dma_slot), 16, (int)b_thr0, (int)(bsoff + (unsigned)(K) * b_kstep), 0, 0)
    // The accumulators START at the bias (round 4): a lane's register r of component b holds channel (r & 3) + 8 (r >> 2) + 4 lh of
    // the wave's 32-channel block.  Added in the epilogue instead (LDS table + barrier + 16 ds_read_b128 + 64 adds) the
    // bias was HALF of the forward epilogue: 7.0 k cycles against backward-data's 3.6 k (tools/probe/phase_stamps.py).
    floatx16 acc[4], accn[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[b][r] = 0.f; accn[b][r] = 0.f; }

    const int frow = wm * 32 + lr;                  // tile row this lane reads; band row = frow + tap offset
    const int b_rd0 = wn * 32 + lr;

    // ---- prologue: band 0 in LDS, B tile 0 on its way there (its DMA leads, the band's loads are behind it: their stores wait for both)
    b_prep();                                        // (the B tile needs no row decode: its loads lead)
#pragma unroll
    for (int k = 0; k < BU; ++k) QK_DMA_B1(k, ldsB0);
    a_prep();
#pragma unroll
    for (int r = 0; r < RPT3; ++r) { decode_row(r); QK_DMA_A(r, ldsA0); }
    QK_STAMP(5);
    // Bias -> accumulators, through LDS: one value per thread fetched behind the band's loads, parked in the last rows of band
    // buffer 1 (first written by a DMA issued in sub-step 1 or later, i.e. behind the NEXT barrier) and read back as 16 broadcast
    // ds_read_b128 per lane right behind the prologue's barrier.  (16 vector float4 loads per lane moved 64 KB per workgroup into
    // registers and cost 1.4 k cycles; wave-uniform scalar loads + a move and a select per register 1.2 k.)
    float bias_v = 0.f;
    if (g.has_bias && tid < 4 * BF && (!PAD || j0 + tid % BF < g.J)) bias_v = bias[(tid / BF) * g.J + j0 + tid % BF];
    QK_STAMP(6);
    a_advance_if_more();
    b_advance_if_more();                             // sub-step s fetches tile s + 1 straight into the other buffer
    // (parked in the LAST rows of band buffer 1: their DMA is issued in sub-step 1 or later, behind a barrier every wave passes after reading this)
    static_assert(4 * BF * 4 <= 8 * 8 * 16, "the bias fits the rows of the band's last pass");
    if (g.has_bias && tid < 4 * BF) (reinterpret_cast<float *>(ldsA1 + A_U) - 4 * BF)[tid] = bias_v;
    __builtin_amdgcn_s_waitcnt((7 << 4) | (15 << 8));                  // vmcnt(0): B tile 0 is in LDS (the band's stores above waited for younger loads already)
    __syncthreads();
    if (g.has_bias) {
        const float4 *br = reinterpret_cast<const float4 *>(reinterpret_cast<float *>(ldsA1 + A_U) - 4 * BF) + wn * 8 + (lane >> 5);
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 bb = br[b * (BF / 4) + 2 * gq];
                acc[b][4 * gq] = bb.x; acc[b][4 * gq + 1] = bb.y; acc[b][4 * gq + 2] = bb.z; acc[b][4 * gq + 3] = bb.w;
            }
    }

    QK_STAMP(1);
    static_assert(band_op(RPT3, KIN, 0, 0) != -2, "no staging schedule for this (row passes, inner taps)");
    static_assert(kBandOpsMax + 2 * BU <= 16, "one staging slot after every second MFMA");

    int s = 0;                                      // global sub-step
    // PAD, Q = 16 (mod 32): the upper 16 channels of the last chunk are padding -- its second 16-deep step multiplies zeros and is
    // skipped (where the staging schedule fits the first step's slots: the 4 x 1 tile form the 16-wide layers take)
    constexpr bool KSKIP = PAD && (BU + kBandOpsMax <= 8);
    const bool q_half = KSKIP && g.Qp != g.Q;
    int ckc = 0;                                    // channel chunk of the group being computed
    static_assert(KIN % 2 == 1, "parity of a group's first sub-step = parity of the group");
    // PIPE (round 5, late): the fragment reads are software-pipelined IN PLACE.  Until now a half step (16 MFMAs) began with its 8
    // ds_read_b128 and its first MFMA waited for five of them: ~100 - 130 cycles per 512 that only the OTHER workgroup of the CU could
    // cover -- a workgroup running alone (its neighbour in prologue / epilogue: 31 % of the time by the phase stamps) ran at 62 %.
    // Now the registers of a fragment are refilled for the NEXT half step right behind the fragment's last MFMA: A[a] behind row a
    // (row a is its only user), the B fragments behind the MFMAs of row 3 in the order the next half step's row 0 consumes them.
    // The sub-step's wait + barrier moves in front of row 3 of its second half: behind it the next B tile is in LDS and the B refills
    // of the next sub-step's first half can be issued under the last four MFMAs (the A refills too where the next sub-step opens a
    // new group: the other band object is complete behind this barrier, not earlier).  No extra registers, no lgkmcnt(0) drain at
    // the barrier (everything this wave read from the buffers about to be overwritten has been consumed by MFMAs in front of it).
    constexpr bool PIPE = (QK_BAND_PIPE != 0) && !PAD;
    if constexpr (PIPE) {
        const int tdir = g.b_rev ? -1 : 1;
        const int arow_first = frow + (g.b_rev ? KIN - 1 : 0);
        uint4 A[4], B[4];
#define QK_LD_A(a_, BAND_, AROW_, KS_) A[a_] = (BAND_)[(AROW_) * 8 + ((a_) >> 1) * PL + ((((a_) & 1) * 4 + (KS_) * 2) ^ ((((AROW_) >> 1) & 7) ^ lh))]
#define QK_LD_B(p_, BOBJ_, KS_) B[p_] = (BOBJ_)[b_rd0 + (((KS_) * 2 + lh) * 4 + (p_)) * BF]
#pragma unroll
        for (int a = 0; a < 4; ++a) QK_LD_A(a, ldsA0, arow_first, 0);
#pragma unroll
        for (int p = 0; p < 4; ++p) QK_LD_B(p, ldsB0, 0);
        for (int gi2 = (g.ablate & 4) ? groups : 0; gi2 < groups; gi2 += 2)
#pragma unroll
        for (int gh = 0; gh < 2; ++gh) {
            const int gi = gi2 + gh;
            if (gi >= groups) break;
            const uint4 *band = gh ? ldsA1 : ldsA0;
            const uint4 *nband = gh ? ldsA0 : ldsA1;
            a_prep();
#pragma unroll
            for (int ti = 0; ti < KIN; ++ti, ++s) {
                int arow = arow_first + ti * tdir;
                // (hipcc hoists the fragment addresses of all KIN sub-steps out of the group loop: ~ 25 VGPRs, 254 - 256 in the 5-tap forms and a
                //  4-byte spill outside the loop in the masked 32-filter one.  Made opaque here -- QK_BAND_NO_HOIST_ADDR -- they are recomputed
                //  under the MFMAs: 232 - 236 VGPRs, no spill, but 1.0 - 1.5 % slower on zeros and on data, same box: not the default)
#ifdef QK_BAND_NO_HOIST_ADDR
                asm volatile("" : "+v"(arow));
#endif
                const int arow_n = arow + tdir;                        // (used where ti + 1 < KIN)
                const int rdpar = (gh + ti) & 1;
                b_prep();
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int bi = 0; bi < 4; ++bi) {
                            constexpr unsigned tbl = TBL;
                            // accumulator order: rows 0..2 run b = 3, 2, 0, 1, row 3 runs b = 0, 1, 3, 2 -- row 3 then frees B[3], B[2], B[0], B[1]
                            // in the order row 0 of the next half step asks for them, and no accumulator is touched by two neighbours
                            const int b = a == 3 ? ((bi & 2) ? (bi ^ 1) : bi) : ((bi & 2) ? (bi & 1) : 3 - bi);
                            if (ks == 1 && a == 3 && bi == 0) {
                                // every wave's DMA units of the next B tile are IN LDS before anybody passes (all but the A loads issued behind them)
#define QK_DMA_WAIT(TI) case TI: { constexpr int n_ld = band_loads_in(RPT3, KIN, TI < KIN ? TI : 0) - ((!TRIM && band_loads_halo_in(RPT3, KIN, TI < KIN ? TI : 0)) ? 1 : 0); \
                __builtin_amdgcn_s_waitcnt(((2 * n_ld) & 15) | (7 << 4) | (15 << 8) | (((2 * n_ld) >> 4) << 14)); } break;       /* vmcnt(2 n_ld) */
                                if (QK_PROBE_ON(16)) {
                                switch (ti) { QK_DMA_WAIT(0) QK_DMA_WAIT(1) QK_DMA_WAIT(2) QK_DMA_WAIT(3) QK_DMA_WAIT(4) default: __builtin_amdgcn_s_waitcnt(0); }
                                __builtin_amdgcn_s_barrier();
                                }
#undef QK_DMA_WAIT
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            if ((tbl >> (a * 4 + b)) & 1u) accn[b] = mfma16(T(), B[a ^ b], A[a], accn[b]);
                            else acc[b] = mfma16(T(), B[a ^ b], A[a], acc[b]);
                            const int f = ks * 16 + a * 4 + bi;
                            if (!QK_PROBE_ON(32)) { }
                            else
                            if (f % 2 == 1) {
                                const int op = f / 2 - BU;                   // the B tile's DMA leads: a whole sub-step to arrive
                                if (op < 0) { if (rdpar) QK_DMA_B1(f / 2, ldsB0); else QK_DMA_B1(f / 2, ldsB1); }
                                else
                                if (op < kBandOpsMax) {
                                    const int q = band_op(RPT3, KIN, ti, op);
                                    if (q >= 0 && q < RPT3) { if (gh) QK_DMA_A(q, ldsA0); else QK_DMA_A(q, ldsA1); }
                                }
                            }
                            // ---- refills for the next half step, into the registers this MFMA was the last to read
                            if (ks == 0) {
                                if (bi == 3) QK_LD_A(a, band, arow, 1);
                                if (a == 3) { if (rdpar) QK_LD_B(3 ^ b, ldsB1, 1); else QK_LD_B(3 ^ b, ldsB0, 1); }
                            } else {
                                if (ti + 1 < KIN) { if (bi == 3) QK_LD_A(a, band, arow_n, 0); }
                                else if (a == 3) QK_LD_A(bi, nband, arow_first, 0);           // (behind the barrier: the next group's band)
                                if (a == 3) { if (rdpar) QK_LD_B(3 ^ b, ldsB0, 0); else QK_LD_B(3 ^ b, ldsB1, 0); }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                b_advance_if_more();
            }
            a_advance_if_more();
        }
#undef QK_LD_A
#undef QK_LD_B
    } else
    for (int gi2 = (g.ablate & 4) ? groups : 0; gi2 < groups; gi2 += 2)
#pragma unroll
    for (int gh = 0; gh < 2; ++gh) {                 // two groups per trip: the B buffer a sub-step reads is then a compile-time OBJECT
        int gi = gi2 + gh;
        if (gi >= groups) break;
        const uint4 *band = gh ? ldsA1 : ldsA0;          // (gi2 is even: the parity of a group is gh, and both band OBJECTS are compile-time constants)
        const bool khalf = KSKIP && q_half && ckc == nkc - 1;
        if (KSKIP) { if (++ckc == nkc) ckc = 0; }
        a_prep();
#pragma unroll
        for (int ti = 0; ti < KIN; ++ti, ++s) {
            const int toff = g.b_rev ? KIN - 1 - ti : ti;
            const int arow = frow + toff;
            const uint4 *a_rd = band + arow * 8;
            const int fsw = ((arow >> 1) & 7) ^ lh;
            const int rdpar = (gh + ti) & 1;                       // (a constant once both loops are unrolled)
            const uint4 *b_rd = (rdpar ? ldsB1 : ldsB0) + b_rd0;
            b_prep();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (KSKIP && ks == 1 && khalf) continue;
                uint4 A[4], B[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) A[a] = a_rd[(a >> 1) * PL + (((a & 1) * 4 + ks * 2) ^ fsw)];
#pragma unroll
                for (int p = 0; p < 4; ++p) B[p] = b_rd[((ks * 2 + lh) * 4 + p) * BF];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        constexpr unsigned tbl = TBL;
                        // kernel fragment as the MFMA's A operand, activation fragment as its B operand (both hold 8
                        // consecutive K values of line `lr`: the roles swap for free): the accumulator tile comes out
                        // TRANSPOSED -- lane = output row, register r = channel (r & 3) + 8 (r >> 2) + 4 lh -- so a
                        // lane's values are contiguous channels of ONE row and the epilogue needs no LDS transpose
                        if ((tbl >> (a * 4 + b)) & 1u) accn[b] = mfma16(T(), B[a ^ b], A[a], accn[b]);
                        else acc[b] = mfma16(T(), B[a ^ b], A[a], acc[b]);
                        const int f = ks * 16 + a * 4 + b;
                        if (f % 2 == 1) {
                            const int op = f / 2 - BU;                   // the B tile's DMA leads: a whole sub-step to arrive
                            if (op < 0) { if (rdpar) QK_DMA_B1(f / 2, ldsB0); else QK_DMA_B1(f / 2, ldsB1); }
                            else
                            if (op < kBandOpsMax) {
                                const int q = band_op(RPT3, KIN, ti, op);      // (the schedule's "store" codes are no-ops: a DMA lands by itself)
                                if (q >= 0 && q < RPT3) { if (gh) QK_DMA_A(q, ldsA0); else QK_DMA_A(q, ldsA1); }
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
            }
            b_advance_if_more();
            // every wave's DMA units of the next tile are IN LDS before anybody passes the barrier: all but the A loads issued behind
            // them (two per row pass loaded in this sub-step; the halo pass, wave 0 only, not counted: wave 0 then waits for two more)
#define QK_DMA_WAIT(TI) case TI: { constexpr int n_ld = band_loads_in(RPT3, KIN, TI < KIN ? TI : 0) - ((!TRIM && band_loads_halo_in(RPT3, KIN, TI < KIN ? TI : 0)) ? 1 : 0); \
                __builtin_amdgcn_s_waitcnt(((2 * n_ld) & 15) | (7 << 4) | (0 << 8) | (((2 * n_ld) >> 4) << 14)); } break;       /* vmcnt(2 n_ld) lgkmcnt(0) */
            switch (ti) { QK_DMA_WAIT(0) QK_DMA_WAIT(1) QK_DMA_WAIT(2) QK_DMA_WAIT(3) QK_DMA_WAIT(4) default: __builtin_amdgcn_s_waitcnt(0); }
#undef QK_DMA_WAIT
            // the bare barrier: __syncthreads() is a workgroup FENCE too, and with band DMAs in flight that fence is vmcnt(0) -- the next
            // band would have to arrive within the sub-step that asked for it.  What the barrier has to order is all here: this wave's
            // fragment reads have returned (lgkmcnt(0)), its units of the next B tile are in LDS (the vmcnt above); the band in flight is
            // complete behind the group's LAST wait (no loads in that sub-step: vmcnt(0)), one barrier before anybody reads it.
            __builtin_amdgcn_s_barrier();
        }
        a_advance_if_more();
    }

    // ---- epilogue: bias + activation (+ chain mask / post-op), 16-byte stores -----------------------------
    // A lane owns output row `lr` of the wave tile and, per component, channels 8g + 4lh + (0..3), g = 0..3.  One
    // v_permlane32_swap per packed dword pairs the lh halves: lanes 0..31 end up with channels 16q .. 16q+7, lanes
    // 32..63 with 16q+8 .. 16q+15 of their row -- two 16-byte stores per component and no trip through LDS (the
    // per-wave transpose patches were 8 - 18 % of these kernels by ablation).
    QK_STAMP(2);
    if ((g.ablate & 8) && acc[0][0] != 123.456f) return;              // (ablate 8: profiling, no epilogue)
    const bool post_on = (EPM || POSTF) && g.post.kind != 0;
    const PostOp psd = (EPM || POSTF) ? resolve_seed(g.post) : g.post;
    const int tr = wm * 32 + lr;                                     // row inside the tile
    bool o_ok;
    long long o_row;                                                 // element offset of the lane's first piece, component 0
    int a_key = 0;                                                   // post-op: alpha index of the lane's row
    {
        const int P = p0 + tr;
        const int line = fastdiv(P, g.dv_mul[0], g.dv_shr[0]);
        const int u = P - line * WP;
        o_ok = line < g.b_nlines && u < g.osp[2] && (!TRIM || tr < BMU);
        o_row = (long long)(line * g.osp[2] + u) * (int)g.out_ss + j0 + wn * 32 + lh * 8;
        if ((EPM || POSTF) && post_on && g.post.alpha_sel >= 0 && o_ok) {
            const int l2 = fastdiv(line, g.dv_mul[1], g.dv_shr[1]), o1 = line - l2 * g.osp[1];
            const int nn = fastdiv(l2, g.dv_mul[2], g.dv_shr[2]), o0 = l2 - nn * g.osp[0];
            a_key = g.post.alpha_sel == 0 ? o0 : g.post.alpha_sel == 1 ? o1 : u;
        }
    }
    float a_val = 0.f, dal = 0.f;
    float *aslab = reinterpret_cast<float *>(ldsA1);                                   // 256 d-alpha sums (backward post-op)
    if ((EPM || POSTF) && post_on && g.post.alpha) a_val = g.post.alpha[a_key];
    // (the K loop's last barrier is behind every wave: the tile buffers are free)
    if (EPM && post_on && g.dalpha && tid < 256) aslab[tid] = 0.f;
    if (EPM && post_on && g.dalpha) __syncthreads();
    uint4 em[EPM ? 4 : 1][2];
    if (EPM && g.ep_mask) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                em[b][q] = (o_ok && (!PAD || j0 + wn * 32 + q * 16 + lh * 8 < g.J))
                               ? *reinterpret_cast<const uint4 *>(static_cast<const T *>(g.ep_mask) + o_row + b * g.J + q * 16)
                               : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        unsigned pk[4][2];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[b][4 * gq + e] - accn[b][4 * gq + e];       // (the bias went in with the first MFMA)
            if (g.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
            pk[gq][0] = pack2(T(), v[0], v[1]);
            pk[gq][1] = pack2(T(), v[2], v[3]);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * q][0], pk[2 * q + 1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * q][1], pk[2 * q + 1][1], false, false);
            uint4 v = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            if (o_ok && (!PAD || j0 + wn * 32 + q * 16 + lh * 8 < g.J)) {       // (PAD: a 16-byte piece lies inside or outside J as a whole)
                const long long o = o_row + b * g.J + q * 16;
                if constexpr (EPM) {
                    if (g.ep_mask) {
                        if (post_on) v = post_bwd8<T>(v, em[b][q], a_val, (unsigned)o, psd, dal);
                        else v = mask8(v, em[b][q]);
                    }
                }
                if constexpr (POSTF) {
                    if (post_on) {
                        if (g.pre_out) *reinterpret_cast<uint4 *>(static_cast<T *>(g.pre_out) + o) = v;
                        v = post_fwd8<T>(v, a_val, (unsigned)o, psd);
                    }
                }
                *reinterpret_cast<uint4 *>(out + o) = v;
            }
        }
    }
    if constexpr (EPM) {
        if (post_on && g.dalpha) {                        // d alpha: wave sums by key -> LDS -> one global atomic per key
            wave_add_by_key(dal, a_key, aslab, lane);
            __syncthreads();
            if (tid < g.post.alpha_len && aslab[tid] != 0.f) atomicAdd(g.dalpha + tid, aslab[tid]);
        }
    }
#ifdef QK_PHASE_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
#endif
    QK_STAMP(3);
#undef QK_STAMP
}

// ---------------------------------------------------------------------------------------
// Point form: every produced row gathers ONE row of the input under ONE tap.
//
// That is a 1 x 1 convolution / dense layer (one tap), and the backward-data of a 'valid' convolution whose kernel
// spans axis 0 (the TimeDistributed dense head run as an (F, 1) convolution): dx[n, f, s, :] = dy[n, s, :] (x) W[f].
// Per 128-row tile the implicit GEMM above runs Q / 32 = 2 K steps -- its pipeline prologue, tap bookkeeping and
// 64 KB epilogue are then 90 % of the tile (measured: 15 us per tile for 1.7 us of MFMAs).  This kernel is the
// streaming form of the same arithmetic:
//   * the tap's re-laid-out kernel (32 KB at Q = J = 64) stays in LDS; persistent workgroups walk units
//     (tap, 64-channel column block, 128-row tile) in tap-major order and reload it only when the tap changes;
//   * A fragments go from global memory straight into registers (a row's 16-byte pieces ARE MFMA A operands),
//     and the registers of a fragment are refilled for the NEXT unit as soon as its last MFMA has issued, so the
//     loads of unit u + 1 fly under the MFMAs and the epilogue of unit u -- no barrier in steady state, the two
//     waves of a SIMD drift apart and overlap each other's epilogue;
//   * epilogue as in k_hgemm16_band (operands swapped: transposed accumulator, v_permlane32_swap, 16-byte stores, no LDS),
//     optional bias / relu / epilogue mask / post-op.
// HBM-bound by construction: the head backward-data writes 367 MB for 94 GFLOP.
// ---------------------------------------------------------------------------------------
// WN = 1 (round 5, late): 32-channel column blocks, 256-row tiles -- produced widths that are multiples of 32 but not of 64 (the head's
// backward-data of a start_filter = 16 model: 64 -> 32) ran the general implicit-GEMM kernel at 6 % of peak (0.28 ms for 184 MB).
template <typename T, int NKC, bool EPM, int WN = 2>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_hgemm16_point(const T *__restrict__ in, const uint4 *__restrict__ wq, const float *__restrict__ bias,
                T *__restrict__ out, const GemmGeom g, const int n_tiles, const int n_units, const unsigned in_bytes,
                const unsigned out_bytes)
{
    static_assert(WN == 1 || WN == 2, "column blocks of 32 or 64 channels");
    constexpr int BM = (8 / WN) * 32, BF = WN * 32;
    constexpr int NF = 2 * NKC;                      // 16-channel K slices per component
    constexpr int B_U = NKC * 16 * BF;               // 16-byte units of one (tap, column block) kernel slice
    constexpr unsigned TBL = kSignConj;              // go16 folds the plain table into the kernel
    __shared__ __attribute__((aligned(16))) uint4 lds[B_U + 4 * BF / 4 + 64];   // kernel slice, its bias (4 x BF floats), 256 d-alpha sums

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    const int JB = g.J / BF;
    const int S = g.osp[1] * g.osp[2];               // rows per (sample, tap)
    const int R = g.batch * S;                       // gathered rows
    const int T_ = g.ks[0];
    const __amdgpu_buffer_rsrc_t rin = make_rsrc16(in, in_bytes);

    // this workgroup's contiguous range of units, unit = (tap * JB + jb) * n_tiles + tile
    const int u_lo = (int)((long long)blockIdx.x * n_units / gridDim.x);
    const int u_hi = (int)((long long)(blockIdx.x + 1) * n_units / gridDim.x);
    if (u_lo >= u_hi) return;

    // byte offset of gathered row r (its component 0, channel 0); rows past the end read as zeros
    auto row_voff = [&](int r) -> unsigned {
        if (r >= R) return kOutOfRange16;
        const int q2 = fastdiv(r, g.dv_mul[0], g.dv_shr[0]), o2 = r - q2 * g.osp[2];
        const int n = fastdiv(q2, g.dv_mul[1], g.dv_shr[1]), o1 = q2 - n * g.osp[1];
        return (unsigned)(n * (int)g.in_sn + o1 * (int)g.in_ss[1] + o2 * (int)g.in_ss[2]) * 2u + (unsigned)lh * 16u;
    };
    uint4 A[4][NF];
    auto load_frag = [&](unsigned voff, int a, int f) {
        A[a][f] = buf_load16b(rin, voff, (unsigned)(a * g.Q + f * 16) * 2u);
    };
    {
        const int tile0 = u_lo % n_tiles;
        const unsigned v0 = row_voff(tile0 * BM + wm * 32 + lr);
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int a = 0; a < 4; ++a) load_frag(v0, a, f);
    }
    int cur_slice = -1;
    // backward post-op (PReLU / dropout derivative of the tensor whose gradient is produced, see k_hgemm16)
    const bool post_on = EPM && g.post.kind != 0;
    const bool post_fwd_relu = !EPM && g.post.kind == 2 && g.post_fwd != 0;
    const PostOp psd = resolve_seed(g.post);
    float *aslab = reinterpret_cast<float *>(lds + B_U + BF);
    if (post_on && g.dalpha && tid < 256) aslab[tid] = 0.f;      // (the first unit's slice load brings the barrier)
    const uint4 *w_rd = lds + wn * 32 + lr;
    const float *bias_q = reinterpret_cast<const float *>(lds + B_U) + wn * 32 + lh * 4;       // + b BF + 8 g: four channels of this lane
    // Stores and mask loads go through buffer resources, rows past the end get an out-of-range offset: no branch
    // around them.
    const __amdgpu_buffer_rsrc_t rout = make_rsrc16(out, out_bytes);
    const __amdgpu_buffer_rsrc_t rmask = make_rsrc16(EPM ? g.ep_mask : out, out_bytes);

    for (int u = u_lo; u < u_hi; ++u) {
        const int slice = u / n_tiles, tile = u - slice * n_tiles;      // slice = tap * JB + jb
        const int tap = slice / JB, jb = slice - tap * JB;
        if (slice != cur_slice) {                    // (uniform over the workgroup: every wave walks the same units)
            __syncthreads();
            const uint4 *src = wq + (long long)tap * (NKC * 16) * g.J + jb * BF;
            uint4 t[B_U / 512];                       // all loads first: one round trip, not B_U / 512 of them
#pragma unroll
            for (int k = 0; k < B_U / 512; ++k) t[k] = src[((tid + k * 512) / BF) * g.J + (tid + k * 512) % BF];
#pragma unroll
            for (int k = 0; k < B_U / 512; ++k) lds[tid + k * 512] = t[k];
            if (g.has_bias && tid < 4 * BF)
                reinterpret_cast<float *>(lds + B_U)[tid] = bias[(tid / BF) * g.J + jb * BF + tid % BF];
            __syncthreads();
            cur_slice = slice;
        }
        // Round 5: the kernel fragment is the MFMA's A operand, the activation fragment its B operand (both hold 8 consecutive K
        // values of line `lr`: the roles swap for free), as in k_hgemm16_band -- the accumulator comes out TRANSPOSED: lane = output
        // row lr of the wave tile, register r = channel (r & 3) + 8 (r >> 2) + 4 lh of the wave's 32-channel block.  The epilogue
        // then needs no trip through LDS (it was 32 two-byte LDS writes + 2 reads + their address VALU per component: the kernel
        // issued ~2000 instructions per wave and unit around 64 MFMAs and was instruction-issue bound, not HBM-bound), the relu
        // form's 1 / (1 - rate) is applied to the fp32 values and its mask to packed pairs.
        unsigned o_off;                                  // byte offset of this lane's row, component 0, channel jb BF + wn 32 + lh 8
        int a_key = 0;
        float a_val = 0.f, dal = 0.f;
        {
            const int r = tile * BM + wm * 32 + lr;
            const int n = fastdiv(r, g.dv_mul[2], g.dv_shr[2]), sp = r - n * S;
            const unsigned o = (unsigned)(((n * T_ + tap) * S + sp) * (int)g.out_ss + jb * BF + wn * 32 + lh * 8) * 2u;
            o_off = r < R ? o : kOutOfRange16;
            if (post_on) {
                const int o1 = fastdiv(sp, g.dv_mul[0], g.dv_shr[0]);
                a_key = g.post.alpha_sel < 0 ? 0 : g.post.alpha_sel == 0 ? tap : g.post.alpha_sel == 1 ? o1 : sp - o1 * g.osp[2];
                a_val = g.post.alpha ? g.post.alpha[a_key] : 0.f;
            }
        }
        // next unit's rows (same rows when only the tap changes); past the range: nothing is fetched
        const bool more = u + 1 < u_hi;
        const int ntile = (u + 1) % n_tiles;
        const unsigned vnext = more ? row_voff(ntile * BM + wm * 32 + lr) : kOutOfRange16;

        uint4 em[EPM ? 4 : 1][2];
        if constexpr (EPM) {
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int q = 0; q < 2; ++q) em[b][q] = buf_load16b(rmask, (g.ablate & 16) ? kOutOfRange16 : o_off, (unsigned)(b * g.J + q * 16) * 2u);
        }
        floatx16 acc[4], accn[4];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            uint4 W[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) W[p] = w_rd[(((f >> 1) * 4 + (f & 1) * 2 + lh) * 4 + p) * BF];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    constexpr unsigned tbl = TBL;
                    const bool ng = (tbl >> (a * 4 + b)) & 1u;
                    // first product of an accumulator starts from zero (no per-unit clearing pass)
                    bool first = f == 0;
#pragma unroll
                    for (int a2 = 0; a2 < 4; ++a2)
                        if (a2 < a && (((tbl >> (a2 * 4 + b)) & 1u) != 0) == ng) first = false;
                    floatx16 zero;
#pragma unroll
                    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
                    if (g.ablate & 4) { if (first) { if (ng) accn[b] = zero; else acc[b] = zero; } }       // (profiling: no MFMAs)
                    else if (ng) accn[b] = mfma16(T(), W[a ^ b], A[a][f], first ? zero : accn[b]);
                    else acc[b] = mfma16(T(), W[a ^ b], A[a][f], first ? zero : acc[b]);
                }
                if (!(g.ablate & 32)) load_frag(vnext, a, f);               // this fragment's registers are free: fetch the next unit's
            }
        }
        // ---- epilogue: a lane owns row lr and, per component, channels 8 g + 4 lh + (0..3), g = 0..3; one v_permlane32_swap per
        //      packed dword pairs the lh halves into 16-byte pieces (channels 16 q + 8 lh .. + 7), two stores per component
        const bool relu_bwd = EPM && post_on && g.post.kind == 2;          // d pre = dy / (1 - rate) where y > 0
        const float pscale = relu_bwd ? g.post.drop_scale : 1.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            constexpr unsigned tbl = TBL;
            constexpr unsigned col_neg = (tbl >> 0 | tbl >> 4 | tbl >> 8 | tbl >> 12) & 0xfu;   // columns with negative entries
            unsigned pk[4][2];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                float v[4];
                float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
                if (g.has_bias) bb = *reinterpret_cast<const float4 *>(bias_q + b * BF + 8 * gq);
                const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[b][4 * gq + e] + bv[e];
                    if ((col_neg >> b) & 1u) v[e] -= accn[b][4 * gq + e];
                    if (g.relu) v[e] = v[e] > 0.f ? v[e] : 0.f;
                    if (EPM) v[e] *= pscale;
                }
                pk[gq][0] = pack2(T(), v[0], v[1]);
                pk[gq][1] = pack2(T(), v[2], v[3]);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * q][0], pk[2 * q + 1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * q][1], pk[2 * q + 1][1], false, false);
                uint4 v = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                const unsigned soff = (unsigned)(b * g.J + q * 16) * 2u;
                if constexpr (EPM) {
                    if (relu_bwd) v = mask8(v, em[b][q]);
                    else if (post_on) v = post_bwd8<T>(v, em[b][q], a_val, (o_off + soff) / 2u, psd, dal);
                    else v = mask8(v, em[b][q]);
                } else {
                    if (post_fwd_relu) v = post_fwd8<T>(v, 0.f, (o_off + soff) / 2u, psd);     // y = dropout(relu(pre))
                }
                buf_store16b(rout, (g.ablate & 8) ? kOutOfRange16 : o_off, soff, v);
            }
        }
        if (post_on && g.dalpha) wave_add_by_key(dal, a_key, aslab, lane);
    }
    if (post_on && g.dalpha) {                            // one global atomic per slope and workgroup
        __syncthreads();
        if (tid < g.post.alpha_len && aslab[tid] != 0.f) atomicAdd(g.dalpha + tid, aslab[tid]);
    }
}

// shapes the point form takes (see k_hgemm16_point), normalised into *o; everything else stays with the implicit-GEMM
// kernels.  A one-tap layer over a dense channels-last tensor (1 x 1 convolution, dense) becomes batch = rows, no
// spatial axes.
inline bool point_geom(const GemmGeom &g, GemmGeom *o)
{
    if (g.has_mask || (g.Q != 32 && g.Q != 64) || g.J % 32 != 0) return false;
    if (g.post.kind != 0) {
        // the backward form (derivative in the epilogue, pre-activation / y in ep_mask), or -- round 4 -- the forward RELU form
        // y = dropout(relu(pre)) with its single output tensor (the TimeDistributed dense layers of the TIMIT model as chain links)
        const bool fwd_relu = g.post_fwd && g.post.kind == 2 && !g.pre_out && !g.ep_mask;
        if (!fwd_relu && (!g.ep_mask || g.post_fwd || g.post.alpha_len > 256 || (g.taps == 1 && g.post.alpha_sel >= 0))) return false;
    }
    *o = g;
    if (g.taps == 1) {
        for (int i = 0; i < 3; ++i)
            if (g.pa[i] != 1 || g.pc[i] != 0 || g.isp[i] != g.osp[i]) return false;
        if (g.in_ss[1] != g.isp[2] * g.in_ss[2] || g.in_ss[0] != g.isp[1] * g.in_ss[1] || g.in_sn != g.isp[0] * g.in_ss[0]) return false;
        o->batch = g.M;
        o->in_sn = g.in_ss[2];
        for (int i = 0; i < 3; ++i) { o->osp[i] = o->isp[i] = 1; o->in_ss[i] = 0; }
    } else {
        if (g.ks[1] != 1 || g.ks[2] != 1 || g.isp[0] != 1 || g.ks[0] != g.osp[0]) return false;
        for (int i = 1; i < 3; ++i)
            if (g.pa[i] != 1 || g.pc[i] != 0 || g.isp[i] != g.osp[i]) return false;
        if (g.pc[0] != 0 || g.pa[0] != 1 || g.pb[0] != -1) return false;      // tap of output position o0 is o0 itself
    }
    if ((long long)o->batch * o->in_sn * 2 >= (1ll << 31) || (long long)g.M * g.out_ss >= (1ll << 31)) return false;
    return true;
}

template <typename T>
int run16_point(const T *in, const uint4 *wq, const float *bias, T *out, GemmGeom g, hipStream_t stream)
{
    const int S = g.osp[1] * g.osp[2], R = g.batch * S;
    fastdiv_of((unsigned)g.osp[2], &g.dv_mul[0], &g.dv_shr[0]);
    fastdiv_of((unsigned)g.osp[1], &g.dv_mul[1], &g.dv_shr[1]);
    fastdiv_of((unsigned)S, &g.dv_mul[2], &g.dv_shr[2]);
    const bool wide = g.J % 64 == 0;                  // 64-channel column blocks x 128-row tiles, or 32 x 256
    const int BM = wide ? 128 : 256, BF = wide ? 64 : 32;
    const int n_tiles = (R + BM - 1) / BM;
    const int n_units = g.ks[0] * (g.J / BF) * n_tiles;
    int blocks = device_cu_count();                   // 8 waves at a 256-register budget: one workgroup per CU
    if (blocks > n_units) blocks = n_units;
    const unsigned in_bytes = (unsigned)((long long)g.batch * g.in_sn * 2);
    const unsigned out_bytes = (unsigned)((long long)g.M * g.out_ss * 2);
#define QK_GO(NKC, E, W) hipLaunchKernelGGL((k_hgemm16_point<T, NKC, E, W>), dim3(blocks), dim3(512), 0, stream, in, wq, bias, out, g, n_tiles, n_units, in_bytes, out_bytes)
    const bool epm = g.ep_mask != nullptr;
    if (wide) {
        if (g.Q == 64) { if (epm) QK_GO(2, true, 2); else QK_GO(2, false, 2); }
        else           { if (epm) QK_GO(1, true, 2); else QK_GO(1, false, 2); }
    } else {
        if (g.Q == 64) { if (epm) QK_GO(2, true, 1); else QK_GO(2, false, 1); }
        else           { if (epm) QK_GO(1, true, 1); else QK_GO(1, false, 1); }
    }
#undef QK_GO
    return hipGetLastError() == hipSuccess ? 1 : QK_ERR_LAUNCH;
}

template <typename T, int WM, int WN, int KIN, bool TRIM, bool PAD = false>
int run16_band(const T *in, const uint4 *wq, const T *zero_line, const float *bias, T *out, const GemmGeom &g,
               hipStream_t stream)
{
    constexpr int BM = WM * 32, BF = WN * 32, BMU = TRIM ? BM - (KIN - 1) : BM, NTHR = WM * WN * 64;
    const int n_mt = (int)(((long long)g.b_nlines * g.b_wp + BMU - 1) / BMU);
    dim3 grid((n_mt + 7) / 8 * 8, (PAD ? g.Jp : g.J) / BF, 1);
    // EPM (epilogue mask, QK_BWD_MASK_DX) is its own instantiation: its eight prefetched mask pieces cost 32 VGPRs
    // ... and so is POSTF (forward post-op: PReLU / dropout, pre-activation written beside y)
#define QK_GO(C, E, P) hipLaunchKernelGGL((k_hgemm16_band<T, WM, WN, KIN, C, TRIM, E, P, PAD>), grid, dim3(NTHR), (g.ablate & 64) ? 40960 : 0, stream, in, wq, zero_line, bias, out, g)      /* (ablate 64: profiling, ONE workgroup per CU) */
    const bool epm = g.ep_mask != nullptr, pf = g.post.kind != 0 && g.post_fwd != 0;
    if (g.sign_tbl != kSignConj) return QK_ERR_LAUNCH;                 // go16 folds the plain table into the kernel
    if (epm) QK_GO(true, true, false); else if (pf) QK_GO(true, false, true); else QK_GO(true, false, false);
#undef QK_GO
    return hipGetLastError() == hipSuccess ? 1 : QK_ERR_LAUNCH;
}

template <typename T, int MT, int WM, int WN>
int run16(const T *in, const T *mask, const uint4 *wq, const T *zero_line, const float *bias, T *out,
          const GemmGeom &g_in, hipStream_t stream)
{
    constexpr int BM = WM * MT * 32, BF = WN * 32;
    const GemmGeom &g = g_in;
    const int n_mt = (g.M + BM - 1) / BM;
    dim3 grid((n_mt + 7) / 8 * 8, g.J / BF, 1);       // padded to the 8 XCDs (see the tile remap)
    const bool m = g.has_mask != 0;
    if (g.sign_tbl != kSignConj) return QK_ERR_LAUNCH;                 // go16 folds the plain table into the kernel
#define QK_GO(C, K) hipLaunchKernelGGL((k_hgemm16<T, MT, WM, WN, C, K>), grid, dim3(512), 0, stream, in, mask, wq, zero_line, bias, out, g)
    if (m) QK_GO(true, true); else QK_GO(true, false);
#undef QK_GO
    return hipGetLastError() == hipSuccess ? 1 : QK_ERR_LAUNCH;
}

template <typename T>
int go16(const void *in, const void *mask, const float *w, const float *bias, void *out, const GemmGeom &g_in,
         bool transposed, void *ws, size_t ws_bytes, hipStream_t stream)
{
    GemmGeom g = g_in;
    g.ablate = debug_ablate();
    { size_t nb = 0; g.dbg_ts = debug_buffer(&nb); if (nb < 64u * 65536u) g.dbg_ts = nullptr; }     // (room for 65536 workgroups x 8 words)
    // One sign table for every 16-bit kernel: the plain table is the conjugate one applied to the conjugated kernel
    // quaternion (S_conv[a][b] = S_conj[a][b] * s[a ^ b] with s = (+, -, -, -)), so the i, j, k components are negated
    // while the kernel is re-laid out (exact in bf16 / fp16) and only the CONJ instantiations exist -- the conjugate
    // table never subtracts into output component 0, i.e. it needs 7 accumulator tiles, not 8, which is what lets the
    // 64-row 4-wave band kernel fit two workgroups per CU without spilling.
    const bool neg_ijk = g.sign_tbl == kSignConv;
    g.sign_tbl = kSignConj;
    for (int i = 0; i < 3; ++i) fastdiv_of((unsigned)g.osp[2 - i], &g.dv_mul[i], &g.dv_shr[i]);
    T *wq = static_cast<T *>(ws);
    const int Cq = transposed ? g.J : g.Q, F = transposed ? g.Q : g.J;
    g.Qp = pad32(g.Q); g.Jp = pad32(g.J);
    const bool padded = g.Qp != g.Q || g.Jp != g.J;      // multiples of 16 only: the band kernel's PAD form or nothing (the caller goes on to the fp32-MFMA kernels)
    GemmGeom bg;
    {
        // 16 / 32 channels per component (start_filter = 16 models): the streaming small-channel kernel (qk_hconv16_small.hip).  Such a
        // shape's workspace holds BOTH 16-bit layouts of the kernel, each in a region of its own -- [band layout + zero line][fragment
        // layout of k_hconv16_small] (qk_*_workspace_bytes; qk_conv_prep_kernels writes both) -- so a cached workspace
        // (ws_has_kernel) is valid whichever kernel a call ends up on: PReLU post-ops and the diagnostic switch take the band form,
        // everything else the small kernel, in any order, on the same buffer.  A workspace without room for the second region
        // (a caller that sized it by hand for the band form) runs the band form.
        Small16 sm;
        const size_t band_bytes = (size_t)g.taps * g.Qp * 4 * g.Jp * 2 + 256;
        if (!g.w_ch_major && small16_shape(g, &bg, &sm) && ws_bytes >= band_bytes + small16_region_bytes(sm)) {
            T *wqs = reinterpret_cast<T *>(static_cast<char *>(ws) + band_bytes);
            if (!g.w_prepped) {                       // a cache miss fills BOTH regions: the caller may flag the buffer as prepped from now on
                if (int rc = launch_prep_small16(std::is_same<T, bf16>::value ? QK_BF16 : QK_F16, w, wqs, Cq, F, transposed ? 1 : 0, neg_ijk ? 1 : 0, sm, stream)) return rc;
                const long long tot = (long long)g.taps * g.Qp * 4 * g.Jp;
                hipLaunchKernelGGL((k_prep_w16<T>), dim3((unsigned)((tot + 255) / 256 > 2048 ? 2048 : (tot + 255) / 256)), dim3(256), 0, stream, w, wq, g.taps, Cq, F, transposed ? 1 : 0, neg_ijk ? 1 : 0, 0);
                if (hipGetLastError() != hipSuccess) return QK_ERR_LAUNCH;
                g.w_prepped = 1;
            }
            const bool post_ok = (g.post.kind == 0 || g.post.kind == 2) && !g.pre_out && !g.dalpha &&
                                 (g.post.kind == 0 || g.post_fwd || g.ep_mask) && !(reinterpret_cast<uintptr_t>(out) & 7) &&
                                 !(reinterpret_cast<uintptr_t>(g.ep_mask) & 7) && !(reinterpret_cast<uintptr_t>(bias) & 15);
            if (post_ok && !(debug_flags() & kDbgNoSmall16)) {
                fastdiv_of((unsigned)bg.b_wp, &bg.dv_mul[0], &bg.dv_shr[0]);
                fastdiv_of((unsigned)bg.osp[1], &bg.dv_mul[1], &bg.dv_shr[1]);
                fastdiv_of((unsigned)bg.osp[0], &bg.dv_mul[2], &bg.dv_shr[2]);
                bg.sign_tbl = kSignConj;
                bg.ablate = g.ablate;
                const int r = launch_hconv16_small(std::is_same<T, bf16>::value ? QK_BF16 : QK_F16, in, wqs, bias, out, bg, sm, stream);
                if (r != 0) { if (r > 0) note_path(QK_PATH_MFMA16_SMALL); return r; }
            }
        }
    }
    if (padded && ((debug_flags() & kDbgNoBand16) || !band_geom(g, 2, &bg))) return 0;
    const long long total = (long long)g.taps * g.Qp * 4 * g.Jp;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (!g.w_prepped) {                              // (the caller vouches for the workspace: qk_conv_desc_t.ws_has_kernel)
        hipLaunchKernelGGL((k_prep_w16<T>), dim3(blocks), dim3(256), 0, stream, w, wq, g.taps, Cq, F, transposed ? 1 : 0, neg_ijk ? 1 : 0, g.w_ch_major);
        if (hipGetLastError() != hipSuccess) return QK_ERR_LAUNCH;
    }
    const uint4 *wq4 = reinterpret_cast<const uint4 *>(wq);
    const T *zero_line = wq + total;                 // 256 zeroed bytes behind the re-laid-out kernel
    if (!padded && !(debug_flags() & kDbgNoPoint16) && point_geom(g, &bg)) {
        note_path(QK_PATH_MFMA16_POINT);
        return run16_point<T>((const T *)in, wq4, bias, (T *)out, bg, stream);
    }
    if (!(debug_flags() & kDbgNoBand16) && band_geom(g, 2, &bg)) {
        fastdiv_of((unsigned)bg.b_wp, &bg.dv_mul[0], &bg.dv_shr[0]);
        fastdiv_of((unsigned)bg.osp[1], &bg.dv_mul[1], &bg.dv_shr[1]);
        fastdiv_of((unsigned)bg.osp[0], &bg.dv_mul[2], &bg.dv_shr[2]);
        const T *ip = (const T *)in;
        T *op = (T *)out;
        note_path(QK_PATH_MFMA16_BAND);
        // Workgroup shape, by measurement on MI355X (tools/gpu_ab.sh, B = 256 TIMIT layers, us per launch):
        //   N = 128 (J = 32):  8 waves 384 / 376 (fwd / bwd-data)  ->  4 waves x 2 per CU, trimmed band  327 / 319
        //   N = 256 (J = 64):  8 waves 1166 / 1158                ->  4 waves (64-row tiles)
        // QK_DBG_BAND16_8WAVES forces the 8-wave tilings everywhere (A/B switch).
        const bool w8 = (debug_flags() & kDbgBand8Waves) != 0;
        const bool w8_wide = w8;
        if (padded) {                                 // (4-wave tiles only: 64 x 256 when the padded width allows, else the trimmed 124 x 128)
            if (bg.ks[2] == 5) return g.Jp % 64 == 0 ? run16_band<T, 2, 2, 5, false, true>(ip, wq4, zero_line, bias, op, bg, stream)
                                                      : run16_band<T, 4, 1, 5, true, true>(ip, wq4, zero_line, bias, op, bg, stream);
            return g.Jp % 64 == 0 ? run16_band<T, 2, 2, 3, false, true>(ip, wq4, zero_line, bias, op, bg, stream)
                                  : run16_band<T, 4, 1, 3, true, true>(ip, wq4, zero_line, bias, op, bg, stream);
        }
        if (bg.ks[2] == 5) {
            if (g.J % 64 == 0) return w8_wide ? run16_band<T, 4, 2, 5, false>(ip, wq4, zero_line, bias, op, bg, stream)
                                              : run16_band<T, 2, 2, 5, false>(ip, wq4, zero_line, bias, op, bg, stream);
            return w8 ? run16_band<T, 8, 1, 5, false>(ip, wq4, zero_line, bias, op, bg, stream)
                      : run16_band<T, 4, 1, 5, true>(ip, wq4, zero_line, bias, op, bg, stream);
        }
        if (g.J % 64 == 0) return w8_wide ? run16_band<T, 4, 2, 3, false>(ip, wq4, zero_line, bias, op, bg, stream)
                                          : run16_band<T, 2, 2, 3, false>(ip, wq4, zero_line, bias, op, bg, stream);
        return w8 ? run16_band<T, 8, 1, 3, false>(ip, wq4, zero_line, bias, op, bg, stream)
                  : run16_band<T, 4, 1, 3, true>(ip, wq4, zero_line, bias, op, bg, stream);
    }
    note_path(QK_PATH_MFMA16);
    if (g.isp[0] == 1 && g.ks[0] > 1 && g.batch > 1 && g.osp[0] > 1) {       // see GemmGeom::o0_major
        g.o0_major = g.batch;
        fastdiv_of((unsigned)g.batch, &g.dv_mul[2], &g.dv_shr[2]);
    }
    if (g.J % 64 == 0) {
        // (256-row tiles, MT = 2, were measured: no gain over 128 rows, and they spill)
        return run16<T, 1, 4, 2>((const T *)in, (const T *)mask, wq4, zero_line, bias, (T *)out, g, stream);
    }
    return run16<T, 1, 8, 1>((const T *)in, (const T *)mask, wq4, zero_line, bias, (T *)out, g, stream);
}

}  // namespace

int launch_prep_w16_batch(int dtype, const PrepJobs &jobs, int n, hipStream_t stream)
{
    if (n <= 0) return 0;
    long long most = 0;
    for (int i = 0; i < n; ++i) {
        const long long t = (long long)jobs.j[i].taps * jobs.j[i].cq * 4 * jobs.j[i].fq;
        if (t > most) most = t;
    }
    int blocks = (int)((most + 255) / 256);
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    dim3 grid((unsigned)blocks, (unsigned)n, 1);
    if (dtype == QK_BF16) hipLaunchKernelGGL((k_prep_w16_batch<bf16>), grid, dim3(256), 0, stream, jobs);
    else if (dtype == QK_F16) hipLaunchKernelGGL((k_prep_w16_batch<f16>), grid, dim3(256), 0, stream, jobs);
    else return QK_ERR_INVALID_ARG;
    return hipGetLastError() == hipSuccess ? 0 : QK_ERR_LAUNCH;
}

// Returns 1 when the 16-bit MFMA path took the call, 0 when the shape is outside its fast path
// (the caller then runs the general fp32-MFMA kernel), < 0 on error.
int try_hgemm_16(int dtype, const void *in, const void *mask, const float *w_f32, const float *bias,
                 void *out, const GemmGeom &g, bool w_is_transposed, void *ws, size_t ws_bytes,
                 hipStream_t stream)
{
    if (dtype != QK_BF16 && dtype != QK_F16) return 0;
    if (g.in_sc != 1 || g.out_sc != 1) return 0;                       // channels_last buffers only
    const long long S = (long long)g.osp[0] * g.osp[1] * g.osp[2];
    if (g.out_sn != S * g.out_ss) return 0;
    if (g.Q % 16 != 0 || g.J % 16 != 0) return 0;                      // (multiples of 16 but not 32: zero-padded band form, go16)
    if (g.taps > 32 || g.pd[0] != 1 || g.pd[1] != 1 || g.pd[2] != 1) return 0;   // tap bit mask; unit-stride map
    const size_t need = (size_t)g.taps * pad32(g.Q) * 4 * pad32(g.J) * 2 + 256;
    if (!ws || ws_bytes < need) return 0;
    if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(ws) | reinterpret_cast<uintptr_t>(mask)) & 15) return 0;
    if (debug_flags() & kDbgNoMfma16) return 0;                        // diagnostic switch
    if (dtype == QK_BF16) return go16<bf16>(in, mask, w_f32, bias, out, g, w_is_transposed, ws, ws_bytes, stream);
    return go16<f16>(in, mask, w_f32, bias, out, g, w_is_transposed, ws, ws_bytes, stream);
}

}  // namespace qk
