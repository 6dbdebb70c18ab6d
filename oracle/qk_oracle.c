/*
 * TEST INFRASTRUCTURE -- CPU restatement of the reference's Hamilton-product hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the CHECKER.  The product path (HIP, csrc/) never links, loads
 * or calls it.
 *
 * Parity status: the reference ships NO tests / golden vectors, so
 * this oracle is pinned against tests/golden/*.npz, which were produced by executing the
 * reference's own layer code (complexnn/conv.py, dense.py, init.py imported from
 * /root/reference) through the torch-float64 keras stand-in (oracle/make_golden.py).
 * tests/test_oracle_golden.py checks every function here against every fixture.
 *
 * What is restated (all arithmetic in double):
 *   qko_fwd   -- QuaternionConv.call  (complexnn/conv.py:288-345): block table
 *                conv.py:327-331, one cross-correlation conv.py:334, bias conv.py:336-341,
 *                activation conv.py:342-343;  and QuaternionDense.call
 *                (complexnn/dense.py:126-164): block table dense.py:139-143 (the TRANSPOSE
 *                of conv's => conj(W) (x) x), matmul dense.py:149, bias/activation
 *                dense.py:159-162.   Dense is rank 0 with conj = 1.
 *   qko_bwd   -- what TF autodiff of those graphs yields for d(input), d(compact kernel),
 *                d(bias) (closed forms: SURVEY.md 8a rows a3 / a9).
 *   qko_fwd_at / qko_dx_at / qko_dw_at / qko_dbias -- the same sums at sampled indices
 *                (float32 activations, OpenMP over the samples): the oracle at BASELINE's
 *                full sizes.  Pinned entry by entry to the same fixtures.
 *
 * Conventions: components r,i,j,k are four contiguous channel blocks;
 * input channel a*Cq+c, output channel b*F+f, compact kernel (*k, Cq, 4F) last axis p*F+f.
 * Padding is passed explicitly as pad_lo (TF 'same'/'causal' offsets are computed by the
 * caller, see oracle.py:tf_pads, restating keras conv_utils / tf.nn.convolution).
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t rank;        /* 0 = dense, 1..3 = conv rank                                  */
    int32_t batch;
    int32_t in_sp[3];    /* input spatial extents, unused trailing dims = 1              */
    int32_t out_sp[3];
    int32_t cq;          /* quaternion input channels  (real channels / 4)               */
    int32_t fq;          /* quaternion filters         (real outputs  / 4)               */
    int32_t kernel[3];
    int32_t stride[3];
    int32_t dil[3];
    int32_t pad_lo[3];
    int32_t ch_first;    /* 1: (N, 4C, *spatial)   0: (N, *spatial, 4C)                  */
    int32_t conj;        /* 0: W (x) x (conv.py:327-331)   1: conj(W) (x) x (dense.py:139-143) */
    int32_t relu;        /* 1: relu, 0: linear                                           */
    int32_t has_bias;
} qko_desc;

/* conv.py:327-331 -- rows = input component a, cols = output component b.
 * entry = sign of compact part (a ^ b):   r:[+r,-i,-j,-k] i:[+i,+r,-k,+j] ...          */
static const int SGN[4][4] = {
    {+1, +1, +1, +1},
    {-1, +1, +1, -1},
    {-1, -1, +1, +1},
    {-1, +1, -1, +1},
};

static inline int sgn(const qko_desc *d, int a, int b) { return d->conj ? SGN[b][a] : SGN[a][b]; }

static inline size_t act_index(const qko_desc *d, int n, size_t s, size_t S, int ch, int C)
{
    return d->ch_first ? ((size_t)n * C + ch) * S + s : ((size_t)n * S + s) * C + ch;
}

/* y = act( hamilton_conv(x, w) + bias ) */
void qko_fwd(const qko_desc *d, const double *x, const double *w, const double *bias, double *y)
{
    const int Cq = d->cq, F = d->fq, Ci = 4 * Cq, Co = 4 * F;
    const size_t Si = (size_t)d->in_sp[0] * d->in_sp[1] * d->in_sp[2];
    const size_t So = (size_t)d->out_sp[0] * d->out_sp[1] * d->out_sp[2];
    for (int n = 0; n < d->batch; ++n)
    for (int o0 = 0; o0 < d->out_sp[0]; ++o0)
    for (int o1 = 0; o1 < d->out_sp[1]; ++o1)
    for (int o2 = 0; o2 < d->out_sp[2]; ++o2) {
        const size_t so = ((size_t)o0 * d->out_sp[1] + o1) * d->out_sp[2] + o2;
        for (int b = 0; b < 4; ++b)
        for (int f = 0; f < F; ++f) {
            double acc = 0.0;
            for (int t0 = 0; t0 < d->kernel[0]; ++t0) {
                const int i0 = o0 * d->stride[0] + t0 * d->dil[0] - d->pad_lo[0];
                if (i0 < 0 || i0 >= d->in_sp[0]) continue;
                for (int t1 = 0; t1 < d->kernel[1]; ++t1) {
                    const int i1 = o1 * d->stride[1] + t1 * d->dil[1] - d->pad_lo[1];
                    if (i1 < 0 || i1 >= d->in_sp[1]) continue;
                    for (int t2 = 0; t2 < d->kernel[2]; ++t2) {
                        const int i2 = o2 * d->stride[2] + t2 * d->dil[2] - d->pad_lo[2];
                        if (i2 < 0 || i2 >= d->in_sp[2]) continue;
                        const size_t si = ((size_t)i0 * d->in_sp[1] + i1) * d->in_sp[2] + i2;
                        const size_t tap = ((size_t)t0 * d->kernel[1] + t1) * d->kernel[2] + t2;
                        for (int a = 0; a < 4; ++a) {
                            const int p = a ^ b;
                            const double s = (double)sgn(d, a, b);
                            for (int c = 0; c < Cq; ++c) {
                                const double xv = x[act_index(d, n, si, Si, a * Cq + c, Ci)];
                                const double wv = w[(tap * Cq + c) * Co + p * F + f];
                                acc += s * xv * wv;
                            }
                        }
                    }
                }
            }
            if (d->has_bias) acc += bias[b * F + f];
            if (d->relu && acc < 0.0) acc = 0.0;
            y[act_index(d, n, so, So, b * F + f, Co)] = acc;
        }
    }
}

/* Gradients of sum(y * dy) w.r.t. x, w, bias.  `y` is the forward output (needed for the
 * relu mask); dx / dw / dbias may each be NULL to skip. */
void qko_bwd(const qko_desc *d, const double *x, const double *w, const double *y,
             const double *dy, double *dx, double *dw, double *dbias)
{
    const int Cq = d->cq, F = d->fq, Ci = 4 * Cq, Co = 4 * F;
    const size_t Si = (size_t)d->in_sp[0] * d->in_sp[1] * d->in_sp[2];
    const size_t So = (size_t)d->out_sp[0] * d->out_sp[1] * d->out_sp[2];
    const size_t taps = (size_t)d->kernel[0] * d->kernel[1] * d->kernel[2];
    if (dx) memset(dx, 0, sizeof(double) * (size_t)d->batch * Si * Ci);
    if (dw) memset(dw, 0, sizeof(double) * taps * Cq * Co);
    if (dbias) memset(dbias, 0, sizeof(double) * Co);
    for (int n = 0; n < d->batch; ++n)
    for (int o0 = 0; o0 < d->out_sp[0]; ++o0)
    for (int o1 = 0; o1 < d->out_sp[1]; ++o1)
    for (int o2 = 0; o2 < d->out_sp[2]; ++o2) {
        const size_t so = ((size_t)o0 * d->out_sp[1] + o1) * d->out_sp[2] + o2;
        for (int b = 0; b < 4; ++b)
        for (int f = 0; f < F; ++f) {
            const size_t yi = act_index(d, n, so, So, b * F + f, Co);
            double g = dy[yi];
            if (d->relu && !(y[yi] > 0.0)) g = 0.0;
            if (g == 0.0) continue;
            if (dbias) dbias[b * F + f] += g;
            for (int t0 = 0; t0 < d->kernel[0]; ++t0) {
                const int i0 = o0 * d->stride[0] + t0 * d->dil[0] - d->pad_lo[0];
                if (i0 < 0 || i0 >= d->in_sp[0]) continue;
                for (int t1 = 0; t1 < d->kernel[1]; ++t1) {
                    const int i1 = o1 * d->stride[1] + t1 * d->dil[1] - d->pad_lo[1];
                    if (i1 < 0 || i1 >= d->in_sp[1]) continue;
                    for (int t2 = 0; t2 < d->kernel[2]; ++t2) {
                        const int i2 = o2 * d->stride[2] + t2 * d->dil[2] - d->pad_lo[2];
                        if (i2 < 0 || i2 >= d->in_sp[2]) continue;
                        const size_t si = ((size_t)i0 * d->in_sp[1] + i1) * d->in_sp[2] + i2;
                        const size_t tap = ((size_t)t0 * d->kernel[1] + t1) * d->kernel[2] + t2;
                        for (int a = 0; a < 4; ++a) {
                            const int p = a ^ b;
                            const double s = (double)sgn(d, a, b) * g;
                            for (int c = 0; c < Cq; ++c) {
                                const size_t xi = act_index(d, n, si, Si, a * Cq + c, Ci);
                                const size_t wi = (tap * Cq + c) * Co + p * F + f;
                                if (dx) dx[xi] += s * w[wi];
                                if (dw) dw[wi] += s * x[xi];
                            }
                        }
                    }
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Sampled entry points: the SAME sums as qko_fwd / qko_bwd, evaluated only at the listed flat
 * indices, so that the HIP path can meet the oracle at BASELINE's full sizes (B = 256: 7e8
 * outputs of 3840 terms each would take hours in full; 4096 of them take milliseconds, and one
 * kernel-gradient entry is a sum over all 716 800 rows).  Activations are float32 here (bf16 /
 * fp16 / fp32 device tensors convert exactly; a float64 copy of a 183 M-element tensor would not
 * fit comfortably), kernel / bias / results are double, every accumulation is double.
 * tests/test_oracle_golden.py pins them to the fixtures (all indices of every fixture).
 * idx[] are flat element indices into y (qko_fwd_at), x (qko_dx_at), the compact kernel (qko_dw_at),
 * each in the tensor's own layout (d->ch_first for activations).
 * ---------------------------------------------------------------------------------------------- */
static inline void act_decode(const qko_desc *d, int64_t flat, size_t S, int C, int *n, size_t *s, int *ch)
{
    if (d->ch_first) { *s = (size_t)(flat % (int64_t)S); flat /= (int64_t)S; *ch = (int)(flat % C); *n = (int)(flat / C); }
    else             { *ch = (int)(flat % C); flat /= C; *s = (size_t)(flat % (int64_t)S); *n = (int)(flat / (int64_t)S); }
}

static inline double masked_g(const qko_desc *d, const float *y, const float *dy, size_t yi)
{
    if (d->relu && !(y[yi] > 0.0f)) return 0.0;
    return (double)dy[yi];
}

void qko_fwd_at(const qko_desc *d, const float *x, const double *w, const double *bias,
                const int64_t *idx, int64_t count, double *out)
{
    const int Cq = d->cq, F = d->fq, Ci = 4 * Cq, Co = 4 * F;
    const size_t Si = (size_t)d->in_sp[0] * d->in_sp[1] * d->in_sp[2];
    const size_t So = (size_t)d->out_sp[0] * d->out_sp[1] * d->out_sp[2];
    #pragma omp parallel for schedule(dynamic, 16)
    for (int64_t q = 0; q < count; ++q) {
        int n, ch; size_t so;
        act_decode(d, idx[q], So, Co, &n, &so, &ch);
        const int b = ch / F, f = ch % F;
        const int o2 = (int)(so % d->out_sp[2]), o1 = (int)((so / d->out_sp[2]) % d->out_sp[1]);
        const int o0 = (int)(so / ((size_t)d->out_sp[2] * d->out_sp[1]));
        double acc = 0.0;
        for (int t0 = 0; t0 < d->kernel[0]; ++t0) {
            const int i0 = o0 * d->stride[0] + t0 * d->dil[0] - d->pad_lo[0];
            if (i0 < 0 || i0 >= d->in_sp[0]) continue;
            for (int t1 = 0; t1 < d->kernel[1]; ++t1) {
                const int i1 = o1 * d->stride[1] + t1 * d->dil[1] - d->pad_lo[1];
                if (i1 < 0 || i1 >= d->in_sp[1]) continue;
                for (int t2 = 0; t2 < d->kernel[2]; ++t2) {
                    const int i2 = o2 * d->stride[2] + t2 * d->dil[2] - d->pad_lo[2];
                    if (i2 < 0 || i2 >= d->in_sp[2]) continue;
                    const size_t si = ((size_t)i0 * d->in_sp[1] + i1) * d->in_sp[2] + i2;
                    const size_t tap = ((size_t)t0 * d->kernel[1] + t1) * d->kernel[2] + t2;
                    for (int a = 0; a < 4; ++a) {
                        const int p = a ^ b;
                        const double s = (double)sgn(d, a, b);
                        for (int c = 0; c < Cq; ++c)
                            acc += s * (double)x[act_index(d, n, si, Si, a * Cq + c, Ci)]
                                     * w[(tap * Cq + c) * Co + p * F + f];
                    }
                }
            }
        }
        if (d->has_bias) acc += bias[b * F + f];
        if (d->relu && acc < 0.0) acc = 0.0;
        out[q] = acc;
    }
}

/* d(input) at the listed elements of x.  y may be NULL when d->relu == 0. */
void qko_dx_at(const qko_desc *d, const double *w, const float *y, const float *dy,
               const int64_t *idx, int64_t count, double *out)
{
    const int Cq = d->cq, F = d->fq, Ci = 4 * Cq, Co = 4 * F;
    const size_t Si = (size_t)d->in_sp[0] * d->in_sp[1] * d->in_sp[2];
    const size_t So = (size_t)d->out_sp[0] * d->out_sp[1] * d->out_sp[2];
    #pragma omp parallel for schedule(dynamic, 16)
    for (int64_t q = 0; q < count; ++q) {
        int n, ch; size_t si;
        act_decode(d, idx[q], Si, Ci, &n, &si, &ch);
        const int a = ch / Cq, c = ch % Cq;
        const int i2 = (int)(si % d->in_sp[2]), i1 = (int)((si / d->in_sp[2]) % d->in_sp[1]);
        const int i0 = (int)(si / ((size_t)d->in_sp[2] * d->in_sp[1]));
        double acc = 0.0;
        for (int t0 = 0; t0 < d->kernel[0]; ++t0) {
            const int u0 = i0 + d->pad_lo[0] - t0 * d->dil[0];            /* = o0 * stride0 */
            if (u0 < 0 || u0 % d->stride[0]) continue;
            const int o0 = u0 / d->stride[0];
            if (o0 >= d->out_sp[0]) continue;
            for (int t1 = 0; t1 < d->kernel[1]; ++t1) {
                const int u1 = i1 + d->pad_lo[1] - t1 * d->dil[1];
                if (u1 < 0 || u1 % d->stride[1]) continue;
                const int o1 = u1 / d->stride[1];
                if (o1 >= d->out_sp[1]) continue;
                for (int t2 = 0; t2 < d->kernel[2]; ++t2) {
                    const int u2 = i2 + d->pad_lo[2] - t2 * d->dil[2];
                    if (u2 < 0 || u2 % d->stride[2]) continue;
                    const int o2 = u2 / d->stride[2];
                    if (o2 >= d->out_sp[2]) continue;
                    const size_t so = ((size_t)o0 * d->out_sp[1] + o1) * d->out_sp[2] + o2;
                    const size_t tap = ((size_t)t0 * d->kernel[1] + t1) * d->kernel[2] + t2;
                    for (int b = 0; b < 4; ++b) {
                        const int p = a ^ b;
                        const double s = (double)sgn(d, a, b);
                        for (int f = 0; f < F; ++f) {
                            const double g = masked_g(d, y, dy, act_index(d, n, so, So, b * F + f, Co));
                            acc += s * g * w[(tap * Cq + c) * Co + p * F + f];
                        }
                    }
                }
            }
        }
        out[q] = acc;
    }
}

/* d(compact kernel) at the listed elements of w: each is a sum over every (sample, output position). */
void qko_dw_at(const qko_desc *d, const float *x, const float *y, const float *dy,
               const int64_t *idx, int64_t count, double *out)
{
    const int Cq = d->cq, F = d->fq, Ci = 4 * Cq, Co = 4 * F;
    const size_t Si = (size_t)d->in_sp[0] * d->in_sp[1] * d->in_sp[2];
    const size_t So = (size_t)d->out_sp[0] * d->out_sp[1] * d->out_sp[2];
    #pragma omp parallel for schedule(dynamic, 1)
    for (int64_t q = 0; q < count; ++q) {
        int64_t flat = idx[q];
        const int col = (int)(flat % Co); flat /= Co;
        const int c = (int)(flat % Cq); flat /= Cq;
        const int t2 = (int)(flat % d->kernel[2]); flat /= d->kernel[2];
        const int t1 = (int)(flat % d->kernel[1]);
        const int t0 = (int)(flat / d->kernel[1]);
        const int p = col / F, f = col % F;
        double acc = 0.0;
        for (int n = 0; n < d->batch; ++n)
        for (int o0 = 0; o0 < d->out_sp[0]; ++o0) {
            const int i0 = o0 * d->stride[0] + t0 * d->dil[0] - d->pad_lo[0];
            if (i0 < 0 || i0 >= d->in_sp[0]) continue;
            for (int o1 = 0; o1 < d->out_sp[1]; ++o1) {
                const int i1 = o1 * d->stride[1] + t1 * d->dil[1] - d->pad_lo[1];
                if (i1 < 0 || i1 >= d->in_sp[1]) continue;
                for (int o2 = 0; o2 < d->out_sp[2]; ++o2) {
                    const int i2 = o2 * d->stride[2] + t2 * d->dil[2] - d->pad_lo[2];
                    if (i2 < 0 || i2 >= d->in_sp[2]) continue;
                    const size_t so = ((size_t)o0 * d->out_sp[1] + o1) * d->out_sp[2] + o2;
                    const size_t si = ((size_t)i0 * d->in_sp[1] + i1) * d->in_sp[2] + i2;
                    for (int a = 0; a < 4; ++a) {
                        const int b = a ^ p;
                        const double g = masked_g(d, y, dy, act_index(d, n, so, So, b * F + f, Co));
                        acc += (double)sgn(d, a, b) * g * (double)x[act_index(d, n, si, Si, a * Cq + c, Ci)];
                    }
                }
            }
        }
        out[q] = acc;
    }
}

/* d(bias): all 4F column sums of the (masked) output gradient. */
void qko_dbias(const qko_desc *d, const float *y, const float *dy, double *out)
{
    const int F = d->fq, Co = 4 * F;
    const size_t So = (size_t)d->out_sp[0] * d->out_sp[1] * d->out_sp[2];
    #pragma omp parallel for schedule(dynamic, 1)
    for (int ch = 0; ch < Co; ++ch) {
        double acc = 0.0;
        for (int n = 0; n < d->batch; ++n)
            for (size_t so = 0; so < So; ++so)
                acc += masked_g(d, y, dy, act_index(d, n, so, So, ch, Co));
        out[ch] = acc;
    }
}

int qko_version(void) { return 2; }
