// K.ctc_batch_cost of the TIMIT model (models/interspeech_model.py:37-39,178 of the reference) as ONE kernel:
//
//     cost[b] = -log p(labels_b | y_pred_b)          y_pred: the model's softmax output (B, T, C), blank = C - 1
//
// Keras 2.x (TensorFlow backend) hands log(y_pred + epsilon()) to tf.nn.ctc_loss as LOGITS, and that op normalises them
// again, so the per-frame log-probabilities are  lp[t][c] = u[t][c] - logsumexp_c u[t][.],  u = log(y_pred + 1e-7);
// ctc_merge_repeated = True (standard CTC), frames >= input_length[b] do not count.  The kernel also returns
// d cost[b] / d y_pred (what TF autodiff gives): with q_t(c) the posterior occupancy of class c at frame t,
//     d cost / d u[t][c] = softmax(u[t])[c] - q_t(c),      d cost / d y_pred[t][c] = (d cost / d u[t][c]) / (y_pred[t][c] + 1e-7).
//
// torch's ctc_loss runs the same recursion as three kernels plus ~15 elementwise passes for the log / log_softmax glue:
// 2.3 ms of a 20 ms training step at B = 256, T = 200.  Here ONE workgroup owns a sample: thread s owns state s of the
// extended label sequence (blank, l1, blank, l2, ..., blank: S = 2 L + 1 states), alpha / beta of the previous frame sit in
// LDS (one barrier per frame), alpha of every frame goes to a workspace for the backward sweep, the emission of the NEXT
// frame is fetched before the barrier of the current one.  Latency-bound by construction (T dependent steps): ~0.1 ms.
#include "qk_common.h"
#include <atomic>

namespace qk {
namespace {

constexpr float kNegInf = -1e30f;
constexpr int CTC_THREADS = 256;

__device__ __forceinline__ float lse2(float a, float b)
{
    const float m = fmaxf(a, b);
    if (m <= -1e29f) return kNegInf;
    return m + __logf(__expf(a - m) + __expf(b - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c)
{
    const float m = fmaxf(a, fmaxf(b, c));
    if (m <= -1e29f) return kNegInf;
    return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}

struct CtcGeom { int B, T, C, Lmax, Smax; float eps; int det; };      // det (QK_DBG_DETERMINISTIC): occupancies summed in a fixed order

// LPS: the whole (T, C) log-probability table of the sample fits in LDS (TIMIT: 200 x 62 floats = 50 KB) and is built once, with
// coalesced loads -- every emission of the two sweeps and the gradient then comes from LDS; otherwise emissions are gathered from
// global memory one frame ahead of their use.
template <typename T, bool LPS>
__global__ void __launch_bounds__(CTC_THREADS)
k_ctc(const T *__restrict__ pred, const int *__restrict__ labels, const int *__restrict__ in_len, const int *__restrict__ lab_len,
      float *__restrict__ cost, T *__restrict__ dpred, float *__restrict__ alpha_ws, const CtcGeom g)
{
    extern __shared__ float smem[];
    // LDS: lse[T] | a[2][Smax] | acc[2][C] | qs[2][Smax] | one float (nll)            (labels are read through registers)
    // (acc and qs alternate with the frame's parity: the threads that turn frame t's occupancies into its gradient do so
    //  while the others already add frame t - 1's -- with one table that was a race)
    float *lse = smem;
    float *ab0 = lse + g.T;
    float *acc2 = ab0 + 2 * g.Smax;
    float *qs2 = acc2 + 2 * g.C;
    float *nll_s = qs2 + 2 * g.Smax;
    float *lpt = nll_s + 4;                    // LPS: lp[t][c]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int Tn = min(max(in_len[b], 0), g.T);
    const int Ln = min(max(lab_len[b], 0), g.Lmax);
    const int S = 2 * Ln + 1;
    const int blank = g.C - 1;
    const T *p = pred + (long long)b * g.T * g.C;
    // state of this thread (S <= CTC_THREADS is checked by the launcher)
    const int s = tid;
    const bool live = s < S;
    // (a label outside [0, C - 2] is clamped: never an out-of-range read; TF raises for such input)
    const int cls = live ? ((s & 1) ? min(max(labels[b * g.Lmax + (s >> 1)], 0), g.C - 1) : blank) : blank;
    const bool skip_ok = live && (s & 1) && s >= 3 && labels[b * g.Lmax + (s >> 1)] != labels[b * g.Lmax + (s >> 1) - 1];   // s-2 -> s
    // the beta recursion looks the other way: s -> s + 2 is allowed when state s + 2 is a label different from state s
    const bool skip_fw = live && (s & 1) && s + 2 < S && labels[b * g.Lmax + (s >> 1)] != labels[b * g.Lmax + (s >> 1) + 1];

    // ---- phase 0: per-frame normaliser lse[t] = logsumexp_c log(p + eps) --------------------------------------------
    if constexpr (LPS) {
        for (int e = tid; e < Tn * g.C; e += CTC_THREADS) lpt[e] = __logf(to_f32(p[e]) + g.eps);      // u[t][c], coalesced
        __syncthreads();
    }
    for (int t = tid; t < Tn; t += CTC_THREADS) {
        float m = kNegInf;
        for (int c = 0; c < g.C; ++c) m = fmaxf(m, LPS ? lpt[t * g.C + c] : __logf(to_f32(p[t * g.C + c]) + g.eps));
        float sum = 0.f;
        for (int c = 0; c < g.C; ++c) sum += __expf((LPS ? lpt[t * g.C + c] : __logf(to_f32(p[t * g.C + c]) + g.eps)) - m);
        lse[t] = m + __logf(sum);
    }
    for (int i = tid; i < 2 * g.C; i += CTC_THREADS) acc2[i] = 0.f;      // C may reach CTC_THREADS: both parity tables
    __syncthreads();
    if constexpr (LPS) {
        for (int e = tid; e < Tn * g.C; e += CTC_THREADS) lpt[e] -= lse[e / g.C];
        __syncthreads();
    }
    auto emit = [&](int t) -> float {          // lp[t][class of this state]
        if constexpr (LPS) return lpt[t * g.C + cls];
        else return __logf(to_f32(p[t * g.C + cls]) + g.eps) - lse[t];
    };
    if (Tn == 0) {             // no frames: the empty labelling has probability 1, any other is impossible
        if (tid == 0) cost[b] = S > 1 ? INFINITY : 0.f;
        if (dpred) for (int e = tid; e < g.T * g.C; e += CTC_THREADS) dpred[(long long)b * g.T * g.C + e] = from_f32<T>(0.f);
        return;
    }

    // ---- phase 1: alpha -------------------------------------------------------------------------------------------------
    float *aws = alpha_ws + (long long)b * g.T * g.Smax;
    float e_cur = live ? emit(0) : 0.f;
    {
        const float a = (live && s < 2) ? e_cur : kNegInf;
        if (live) { ab0[s] = a; aws[s] = a; }
    }
    float e_next = (live && Tn > 1) ? emit(1) : 0.f;
    __syncthreads();
    for (int t = 1; t < Tn; ++t) {
        const float *ap = ab0 + ((t - 1) & 1) * g.Smax;
        float *an = ab0 + (t & 1) * g.Smax;
        e_cur = e_next;
        if (live && t + 1 < Tn) e_next = emit(t + 1);              // in flight across the barrier
        if (live) {
            const float a = lse3(ap[s], s >= 1 ? ap[s - 1] : kNegInf, skip_ok ? ap[s - 2] : kNegInf);
            const float v = a <= -1e29f ? kNegInf : a + e_cur;
            an[s] = v;
            aws[(long long)t * g.Smax + s] = v;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float *al = ab0 + ((Tn - 1) & 1) * g.Smax;
        const float ll = lse2(al[S - 1], S >= 2 ? al[S - 2] : kNegInf);
        nll_s[0] = -ll;
        cost[b] = ll <= -1e29f ? INFINITY : -ll;
    }
    __syncthreads();
    if (!dpred) return;
    const float nll = nll_s[0];
    if (nll >= 1e29f) {
        // no alignment fits (more labels, counting the blanks between repeats, than frames): the cost is +inf and has no
        // gradient.  (TensorFlow's ctc_loss raises "Not enough time for target transition sequence" here; a finite
        // softmax / (p + eps) "gradient" with all occupancies zero would push the optimiser somewhere meaningless.)
        for (int e = tid; e < g.T * g.C; e += CTC_THREADS) dpred[(long long)b * g.T * g.C + e] = from_f32<T>(0.f);
        return;
    }

    // ---- phase 2: beta, posterior occupancies, gradient ------------------------------------------------------------------
    T *dp = dpred + (long long)b * g.T * g.C;
    for (int e = Tn * g.C + tid; e < g.T * g.C; e += CTC_THREADS) dp[e] = from_f32<T>(0.f);       // frames past the input length
    e_cur = live ? emit(Tn - 1) : 0.f;
    float a_cur = live ? aws[(long long)(Tn - 1) * g.Smax + s] : kNegInf;
    float bprev = kNegInf;                      // this thread's beta of frame t + 1 lives in LDS (neighbours read it)
    for (int t = Tn - 1; t >= 0; --t) {
        float *bn = ab0 + (t & 1) * g.Smax;
        const float *bp = ab0 + ((t + 1) & 1) * g.Smax;
        float bt = kNegInf;
        if (live) {
            if (t == Tn - 1) bt = (s == S - 1 || s == S - 2) ? e_cur : kNegInf;
            else {
                const float v = lse3(bp[s], s + 1 < S ? bp[s + 1] : kNegInf, skip_fw ? bp[s + 2] : kNegInf);
                bt = v <= -1e29f ? kNegInf : v + e_cur;
            }
            bn[s] = bt;
        }
        // occupancy of this state at frame t: alpha and beta both hold the emission of frame t
        float q = 0.f;
        if (live && a_cur > -1e29f && bt > -1e29f) q = __expf(a_cur + bt - e_cur + nll);
        // blanks (even states) are half of the states: reduce them inside the wave first, one LDS atomic per wave
        float *acc = acc2 + (t & 1) * g.C, *qs = qs2 + (t & 1) * g.Smax;
        if (g.det) {
            if (live) qs[s] = q;                      // summed per class in state order below: no float atomics
        } else {
            float qb = (live && !(s & 1)) ? q : 0.f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) qb += __shfl_xor(qb, o);
            if (lane == 0 && qb != 0.f) atomicAdd(&acc[blank], qb);
            if (live && (s & 1) && q != 0.f) atomicAdd(&acc[cls], q);
        }
        // next frame's operands, in flight across the barriers
        float e_nx = 0.f, a_nx = kNegInf;
        if (live && t > 0) { e_nx = emit(t - 1); a_nx = aws[(long long)(t - 1) * g.Smax + s]; }
        __syncthreads();
        for (int c = tid; c < g.C; c += CTC_THREADS) {
            float pc, soft;
            if constexpr (LPS) { const float l = lpt[t * g.C + c]; soft = __expf(l); pc = __expf(l + lse[t]); }
            else { pc = to_f32(p[t * g.C + c]) + g.eps; soft = __expf(__logf(pc) - lse[t]); }
            float oc = acc[c];
            if (g.det) {
                oc = 0.f;
                for (int st = 0; st < S; ++st) {
                    const int cl = (st & 1) ? min(max(labels[b * g.Lmax + (st >> 1)], 0), g.C - 1) : blank;
                    if (cl == c) oc += qs[st];
                }
            }
            const float gu = soft - oc;
            dp[t * g.C + c] = from_f32<T>(gu / pc);
            acc[c] = 0.f;
        }
        e_cur = e_nx; a_cur = a_nx; bprev = bt;
        __syncthreads();
    }
    (void)bprev;
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 4: the fast form, taken whenever the sample's (T, C) table fits LDS twice (TIMIT: 2 x 50 KB).  k_ctc above walks the
// lattice twice, one after the other (2 T dependent steps), and computes the gradient of a frame INSIDE the backward sweep
// (62 exp / div, the stores and two barriers per frame): 233 us at B = 256, T = 200 -- 1200 cycles per dependent step.  Here
//   * the alpha sweep (threads 0 .. 255) and the beta sweep (threads 256 .. 511) run CONCURRENTLY, frame i and frame
//     Tn - 1 - i in the same step: T dependent steps, one barrier each, nothing but the recursion inside;
//   * both lattices go to the workspace; the posterior occupancies of ALL frames are then formed in parallel (a wave takes
//     64 consecutive states of one frame: blanks reduced in the wave, labels by LDS atomics into an occupancy table
//     occ[t][c]) and the gradient table is written in one coalesced pass.
// Same arithmetic, same results as k_ctc (tests/test_models.py pins both to the Keras restatement).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int CTCF_THREADS = 512;

template <typename T>
__global__ void __launch_bounds__(CTCF_THREADS)
k_ctc_fast(const T *__restrict__ pred, const int *__restrict__ labels, const int *__restrict__ in_len, const int *__restrict__ lab_len,
           float *__restrict__ cost, T *__restrict__ dpred, float *__restrict__ alpha_ws, float *__restrict__ beta_ws, const CtcGeom g)
{
    extern __shared__ float smem[];
    // LDS: lse[T] | aa[2][Smax] | bb[2][Smax] | cls[Smax] (int) | 4 floats | lp[T][C] | occ[T][C]
    float *lse = smem;
    float *aa = lse + g.T;
    float *bb = aa + 2 * g.Smax;
    int *cls_l = reinterpret_cast<int *>(bb + 2 * g.Smax);
    float *nll_s = reinterpret_cast<float *>(cls_l + g.Smax);
    float *lpt = nll_s + 4;
    float *occ = lpt + g.T * g.C;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int Tn = min(max(in_len[b], 0), g.T);
    const int Ln = min(max(lab_len[b], 0), g.Lmax);
    const int S = 2 * Ln + 1;
    const int blank = g.C - 1;
    const T *p = pred + (long long)b * g.T * g.C;
    const bool beta_half = tid >= 256;
    const int s = tid & 255;
    const bool live = s < S;
    const int lab_s = (live && (s & 1)) ? labels[b * g.Lmax + (s >> 1)] : 0;
    const int cls = live ? ((s & 1) ? min(max(lab_s, 0), g.C - 1) : blank) : blank;
    // alpha: s - 2 -> s allowed when state s is a label different from the previous label; beta looks the other way
    const bool skip_bw = live && (s & 1) && s >= 3 && lab_s != labels[b * g.Lmax + (s >> 1) - 1];
    const bool skip_fw = live && (s & 1) && s + 2 < S && lab_s != labels[b * g.Lmax + (s >> 1) + 1];
    if (!beta_half && s < g.Smax) cls_l[s] = cls;

    // ---- phase 0: u = log(p + eps), per-frame normaliser, lp = u - lse; occupancy table cleared ------------------------
    for (int e = tid; e < Tn * g.C; e += CTCF_THREADS) { lpt[e] = __logf(to_f32(p[e]) + g.eps); occ[e] = 0.f; }
    __syncthreads();
    for (int t = tid; t < Tn; t += CTCF_THREADS) {
        float m = kNegInf;
        for (int c = 0; c < g.C; ++c) m = fmaxf(m, lpt[t * g.C + c]);
        float sum = 0.f;
        for (int c = 0; c < g.C; ++c) sum += __expf(lpt[t * g.C + c] - m);
        lse[t] = m + __logf(sum);
    }
    __syncthreads();
    for (int e = tid; e < Tn * g.C; e += CTCF_THREADS) lpt[e] -= lse[e / g.C];
    if (Tn == 0) {             // no frames: the empty labelling has probability 1, any other is impossible
        if (tid == 0) cost[b] = S > 1 ? INFINITY : 0.f;
        if (dpred) for (int e = tid; e < g.T * g.C; e += CTCF_THREADS) dpred[(long long)b * g.T * g.C + e] = from_f32<T>(0.f);
        return;
    }
    __syncthreads();

    // ---- phase 1: both sweeps at once ------------------------------------------------------------------------------------
    float *lat = (beta_half ? beta_ws : alpha_ws) + (long long)b * g.T * g.Smax;
    float *buf = beta_half ? bb : aa;
    for (int i = 0; i < Tn; ++i) {
        const int t = beta_half ? Tn - 1 - i : i;
        const float *prev = buf + ((i + 1) & 1) * g.Smax;
        float v = kNegInf;
        if (live) {
            const float e = lpt[t * g.C + cls];
            if (i == 0) {
                const bool start = beta_half ? (s == S - 1 || s == S - 2) : s < 2;
                v = start ? e : kNegInf;
            } else if (beta_half) {
                const float a = lse3(prev[s], s + 1 < S ? prev[s + 1] : kNegInf, skip_fw ? prev[s + 2] : kNegInf);
                v = a <= -1e29f ? kNegInf : a + e;
            } else {
                const float a = lse3(prev[s], s >= 1 ? prev[s - 1] : kNegInf, skip_bw ? prev[s - 2] : kNegInf);
                v = a <= -1e29f ? kNegInf : a + e;
            }
            buf[(i & 1) * g.Smax + s] = v;
            lat[(long long)t * g.Smax + s] = v;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float *al = aa + ((Tn - 1) & 1) * g.Smax;
        const float ll = lse2(al[S - 1], S >= 2 ? al[S - 2] : kNegInf);
        nll_s[0] = -ll;
        cost[b] = ll <= -1e29f ? INFINITY : -ll;
    }
    __syncthreads();
    if (!dpred) return;
    const float nll = nll_s[0];
    T *dp = dpred + (long long)b * g.T * g.C;
    for (int e = Tn * g.C + tid; e < g.T * g.C; e += CTCF_THREADS) dp[e] = from_f32<T>(0.f);       // frames past the input length
    if (nll >= 1e29f) {        // no alignment fits: +inf cost, no gradient (see k_ctc)
        for (int e = tid; e < Tn * g.C; e += CTCF_THREADS) dp[e] = from_f32<T>(0.f);
        return;
    }

    // ---- phase 2: occupancies of every (frame, state) in parallel, then the gradient table -----------------------------------
    const float *al = alpha_ws + (long long)b * g.T * g.Smax, *be = beta_ws + (long long)b * g.T * g.Smax;
    const int Sp = (S + 63) & ~63;                   // a wave = 64 consecutive states of ONE frame
    if (g.det) {
        // QK_DBG_DETERMINISTIC: one owner per (frame, class) sums its states in order (Tn x C x S work instead of Tn x S)
        for (int e = tid; e < Tn * g.C; e += CTCF_THREADS) {
            const int t = e / g.C, c = e - t * g.C;
            float oc = 0.f;
            for (int st = 0; st < S; ++st) {
                if (cls_l[st] != c) continue;
                const float a = al[(long long)t * g.Smax + st], bt = be[(long long)t * g.Smax + st];
                if (a > -1e29f && bt > -1e29f) oc += __expf(a + bt - lpt[e] + nll);
            }
            occ[e] = oc;
        }
    } else
    for (int idx = tid; idx < Tn * Sp; idx += CTCF_THREADS) {
        const int t = idx / Sp, st = idx - t * Sp;
        float q = 0.f;
        int c = blank;
        if (st < S) {
            c = cls_l[st];
            const float a = al[(long long)t * g.Smax + st], bt = be[(long long)t * g.Smax + st];
            if (a > -1e29f && bt > -1e29f) q = __expf(a + bt - lpt[t * g.C + c] + nll);       // both hold frame t's emission
        }
        float qb = (st & 1) ? 0.f : q;               // blanks: half of the states, one class -- reduce inside the wave
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) qb += __shfl_xor(qb, o);
        if (lane == 0 && qb != 0.f) atomicAdd(&occ[t * g.C + blank], qb);
        if ((st & 1) && q != 0.f) atomicAdd(&occ[t * g.C + c], q);
    }
    __syncthreads();
    for (int e = tid; e < Tn * g.C; e += CTCF_THREADS) {
        const float l = lpt[e];
        const float soft = __expf(l), pc = __expf(l + lse[e / g.C]);
        dp[e] = from_f32<T>((soft - occ[e]) / pc);
    }
}

}  // namespace

size_t ctc_workspace_bytes(int B, int T, int Lmax) { return 2 * (size_t)B * T * (2 * Lmax + 1) * sizeof(float); }   // alpha and beta lattices

int launch_ctc(int dtype, int B, int T, int C, const void *pred, const int *labels, int Lmax, const int *in_len, const int *lab_len,
               float *cost, void *dpred, float *ws, hipStream_t stream)
{
    CtcGeom g;
    g.B = B; g.T = T; g.C = C; g.Lmax = Lmax; g.Smax = 2 * Lmax + 1; g.eps = 1e-7f;
    g.det = (debug_flags() & kDbgDeterministic) ? 1 : 0;
    const size_t base = (size_t)(T + 4 * g.Smax + 2 * C + 4) * sizeof(float);
    const size_t full = base + (size_t)T * C * sizeof(float);
    if (g.Smax > CTC_THREADS || C > CTC_THREADS || base > 64 * 1024) return QK_ERR_UNSUPPORTED;
    // fast form: both tables (log-probabilities, occupancies) of the sample in LDS
    const size_t fast_lds = (size_t)(T + 5 * g.Smax + 4 + 2 * (size_t)T * C) * sizeof(float);
    if (fast_lds <= 150 * 1024 && !(debug_flags() & kDbgCtcTwoSweeps)) {
        float *beta_ws = ws + (size_t)B * T * g.Smax;
        dim3 grid((unsigned)B), block(CTCF_THREADS);
        // More than 64 KB of dynamic LDS must be asked for explicitly.  The largest size granted so far is remembered per DEVICE and
        // element type (the attribute belongs to the device's copy of the function): the attribute call is made only when a call
        // needs more.  A device that refuses the size, or a launch of the fast form that fails, is no error: the two-sweep kernel
        // below takes the call.
        bool fast_ok = true;
        constexpr int kMaxDevices = 64;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) { (void)hipGetLastError(); dev = 0; }
#define QK_CTCF(TT) do { static std::atomic<int> granted[kMaxDevices]; \
        if ((int)fast_lds > granted[dev].load(std::memory_order_relaxed)) { \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ctc_fast<TT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fast_lds) == hipSuccess) \
                granted[dev].store((int)fast_lds, std::memory_order_relaxed); \
            else { (void)hipGetLastError(); fast_ok = false; } } \
        if (fast_ok) { hipLaunchKernelGGL((k_ctc_fast<TT>), grid, block, fast_lds, stream, (const TT *)pred, labels, in_len, lab_len, cost, (TT *)dpred, ws, beta_ws, g); \
            if (hipGetLastError() != hipSuccess) { granted[dev].store(0, std::memory_order_relaxed); fast_ok = false; } } } while (0)
        switch (dtype) {
        case QK_F32: QK_CTCF(float); break;
        case QK_BF16: QK_CTCF(bf16); break;
        case QK_F16: QK_CTCF(f16); break;
        default: return QK_ERR_INVALID_ARG;
        }
#undef QK_CTCF
        if (fast_ok) return 0;
    }
    const bool lps = full <= 64 * 1024;
    const size_t lds = lps ? full : base;
    dim3 grid((unsigned)B), block(CTC_THREADS);
#define QK_CTC(TT, L) hipLaunchKernelGGL((k_ctc<TT, L>), grid, block, lds, stream, (const TT *)pred, labels, in_len, lab_len, cost, (TT *)dpred, ws, g)
    switch (dtype) {
    case QK_F32: if (lps) QK_CTC(float, true); else QK_CTC(float, false); break;
    case QK_BF16: if (lps) QK_CTC(bf16, true); else QK_CTC(bf16, false); break;
    case QK_F16: if (lps) QK_CTC(f16, true); else QK_CTC(f16, false); break;
    default: return QK_ERR_INVALID_ARG;
    }
#undef QK_CTC
    return hipGetLastError() == hipSuccess ? 0 : QK_ERR_LAUNCH;
}

}  // namespace qk
