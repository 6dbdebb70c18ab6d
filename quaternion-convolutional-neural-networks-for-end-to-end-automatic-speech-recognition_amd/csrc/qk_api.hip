// C-ABI front end (include/qk.h): descriptor validation, geometry, kernel selection.
// No device allocation, no host synchronisation, nothing retained after return.
#include <limits.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "qk_common.h"

#include <atomic>
#include <mutex>
#include <vector>

namespace qk {

static thread_local char g_err[512] = "";
static thread_local int g_path = QK_PATH_NONE;
void note_path(int p) { g_path = p; }

namespace {
unsigned env_flags()
{
    unsigned f = 0;
    if (getenv("QK_NO_MFMA16")) f |= kDbgNoMfma16;
    if (getenv("QK_NO_BAND16")) f |= kDbgNoBand16;
    if (getenv("QK_NO_BAND32")) f |= kDbgNoBand32;
    if (getenv("QK_WGRAD16_ONE_TAP")) f |= kDbgWgradOneTap;
    if (getenv("QK_BAND16_8WAVES")) f |= kDbgBand8Waves;
    if (getenv("QK_NO_WGRAD_BAND")) f |= kDbgNoWgradBand;
    if (getenv("QK_NO_POINT16")) f |= kDbgNoPoint16;
    if (getenv("QK_CTC_TWO_SWEEPS")) f |= kDbgCtcTwoSweeps;
    if (getenv("QK_DETERMINISTIC")) f |= kDbgDeterministic;
    if (getenv("QK_WGRAD_BAND_V1")) f |= kDbgWgradBandV1;
    if (getenv("QK_NO_SMALL16")) f |= kDbgNoSmall16;
    // graph-level switches: stored here, acted on by the host-side model code (include/qk.h)
    if (getenv("QK_NO_CONV_CHAIN")) f |= QK_DBG_NO_CONV_CHAIN;
    if (getenv("QK_NO_FUSED_PRELU")) f |= QK_DBG_NO_FUSED_PRELU;
    if (getenv("QK_NO_FUSED_DROPOUT")) f |= QK_DBG_NO_FUSED_DROPOUT;
    if (getenv("QK_NO_FUSED_CTC")) f |= QK_DBG_NO_FUSED_CTC;
    if (getenv("QK_NO_FUSED_FIRST")) f |= QK_DBG_NO_FUSED_FIRST;
    if (getenv("QK_NO_DENSE_IN_CHAIN")) f |= QK_DBG_NO_DENSE_IN_CHAIN;
    if (getenv("QK_NO_FUSED_SOFTMAX")) f |= QK_DBG_NO_FUSED_SOFTMAX;
    if (const char *ab = getenv("QK_ABLATE")) f |= ((unsigned)atoi(ab) << kDbgAblateShift) & kDbgAblateMask;
    return f;
}
std::atomic<unsigned> &dbg_word()
{
    static std::atomic<unsigned> w{env_flags()};          // thread-safe one-time initialisation (C++11)
    return w;
}
struct ForceCfg { bool set; int policy, bq; };
const ForceCfg &force_cfg()
{
    static const ForceCfg fc = [] {
        ForceCfg c = {false, -1, -1};
        if (const char *f = getenv("QK_FORCE_CFG")) c.set = sscanf(f, "%d,%d", &c.policy, &c.bq) == 2;
        return c;
    }();
    return fc;
}
}  // namespace

unsigned debug_flags() { return dbg_word().load(std::memory_order_relaxed); }
static std::atomic<unsigned long long *> g_dbg_buf{nullptr};
static std::atomic<size_t> g_dbg_bytes{0};
unsigned long long *debug_buffer(size_t *bytes) { if (bytes) *bytes = g_dbg_bytes.load(); return g_dbg_buf.load(); }

// ---- per-call timing (qk_prof_*, include/qk.h) ------------------------------------------------------------------
// Off: one relaxed atomic load per compute call.  On: a pair of HIP events around the call's launches ON THE CALLER'S
// STREAM, kept in a process-wide list until the next qk_prof_enable(1); qk_prof_get synchronises the pair it reads.
// This is how bench.py times the kernels INSIDE the training step (autograd runs the backward on another thread and
// several kernels per C call, so neither Python-side events nor per-kernel wrappers can).
namespace {
struct ProfRec { int op, dtype, path; long long rows; int n, k; hipEvent_t e0, e1; };
struct Prof { std::mutex mu; std::atomic<int> on{0}; std::vector<ProfRec> recs; };
Prof &prof() { static Prof p; return p; }
}  // namespace
static thread_local int g_prof_nest = 0;     // a call that re-enters the front end (channels_first re-layout) is timed once, as a whole
struct ProfScope {
    bool live = false;
    ProfRec r;
    hipStream_t stream;
    ProfScope(int op, const qk_conv_desc_t *d, hipStream_t st) : stream(st)
    {
        if (g_prof_nest++ > 0) return;
        if (!prof().on.load(std::memory_order_relaxed) || !d) return;
        r.op = op; r.dtype = d->dtype; r.path = QK_PATH_NONE;
        r.rows = (long long)d->batch * d->out_spatial[0] * d->out_spatial[1] * d->out_spatial[2];
        r.n = 4 * d->fq;
        r.k = 4 * d->cq * d->kernel[0] * d->kernel[1] * d->kernel[2];
        if (hipEventCreate(&r.e0) != hipSuccess) return;
        if (hipEventCreate(&r.e1) != hipSuccess) { (void)hipEventDestroy(r.e0); return; }
        live = hipEventRecord(r.e0, stream) == hipSuccess;
    }
    ~ProfScope()
    {
        --g_prof_nest;
        if (!live) return;
        (void)hipEventRecord(r.e1, stream);
        r.path = g_path;
        std::lock_guard<std::mutex> lk(prof().mu);
        prof().recs.push_back(r);
    }
};
bool debug_force_cfg(int *policy, int *bq)
{
    const ForceCfg &c = force_cfg();
    if (c.set) { *policy = c.policy; *bq = c.bq; }
    return c.set;
}
int device_cu_count()
{
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
        return 256;
    return n;
}

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#define QK_DECL(sfx)                                                                              \
    int launch_hgemm_##sfx(const void *, const void *, const float *, const float *, void *,      \
                           const GemmGeom &, bool, hipStream_t);                                  \
    int launch_wgrad_##sfx(const void *, const void *, const void *, float *, float *,            \
                           const WgradGeom &, bool, hipStream_t);
QK_DECL(f32)
QK_DECL(bf16)
QK_DECL(f16)
#undef QK_DECL

// 16-bit-input MFMA fast path (qk_hgemm_bf16mfma.hip); returns 1 when it took the call
int try_hgemm_16(int dtype, const void *in, const void *mask, const float *w_f32, const float *bias,
                 void *out, const GemmGeom &g, bool w_is_transposed, void *ws, size_t ws_bytes,
                 hipStream_t stream);

// 16-bit-input MFMA backward-weight, band form (qk_wgrad_band_bf16mfma.hip); returns 1 when it took the call
int try_wgrad_band_16(int dtype, const void *x, const void *dy, const void *ymask, float *dw, float *dbias,
                      const WgradGeom &g, hipStream_t stream);
// 16-bit-input MFMA backward-weight (qk_wgrad_bf16mfma.hip); returns 1 when it took the call
int try_wgrad_16(int dtype, const void *x, const void *dy, const void *ymask, float *dw, float *dbias,
                 const WgradGeom &g, hipStream_t stream);

int launch_hgemm(int dtype, const void *in, const void *mask, const float *wk, const float *bias,
                 void *out, const GemmGeom &g, bool vec_ok, hipStream_t stream)
{
    switch (dtype) {
    case QK_F32: return launch_hgemm_f32(in, mask, wk, bias, out, g, vec_ok, stream);
    case QK_BF16: return launch_hgemm_bf16(in, mask, wk, bias, out, g, vec_ok, stream);
    case QK_F16: return launch_hgemm_f16(in, mask, wk, bias, out, g, vec_ok, stream);
    }
    return QK_ERR_INVALID_ARG;
}

int launch_wgrad(int dtype, const void *x, const void *dy, const void *ymask, float *dw,
                 float *dbias, WgradGeom g, bool vec_ok, hipStream_t stream)
{
    switch (dtype) {
    case QK_F32: return launch_wgrad_f32(x, dy, ymask, dw, dbias, g, vec_ok, stream);
    case QK_BF16: return launch_wgrad_bf16(x, dy, ymask, dw, dbias, g, vec_ok, stream);
    case QK_F16: return launch_wgrad_f16(x, dy, ymask, dw, dbias, g, vec_ok, stream);
    }
    return QK_ERR_INVALID_ARG;
}

namespace {

size_t elem_bytes(int dtype) { return dtype == QK_F32 ? 4 : 2; }

bool aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// 4-element vector loads need 16 B (f32) / 8 B (16-bit) alignment
bool vec_aligned(const void *p, int dtype) { return aligned(p, 4 * elem_bytes(dtype)); }

int validate(const qk_conv_desc_t *d, bool allow_rank0)
{
    if (!d) { set_error("descriptor is NULL"); return QK_ERR_INVALID_ARG; }
    if (d->rank < (allow_rank0 ? 0 : 1) || d->rank > 3) {
        set_error("rank %d not in [1,3]", d->rank); return QK_ERR_INVALID_ARG;
    }
    if (d->batch <= 0 || d->cq <= 0 || d->fq <= 0) {
        set_error("batch/cq/fq must be positive (got %d/%d/%d)", d->batch, d->cq, d->fq);
        return QK_ERR_INVALID_ARG;
    }
    long long si = 1, so = 1, taps = 1;
    for (int i = 0; i < 3; ++i) {
        if (d->in_spatial[i] <= 0 || d->out_spatial[i] <= 0 || d->kernel[i] <= 0 ||
            d->stride[i] <= 0 || d->dilation[i] <= 0 || d->pad_lo[i] < 0) {
            set_error("axis %d: extents/kernel/stride/dilation must be > 0 and pad_lo >= 0", i);
            return QK_ERR_INVALID_ARG;
        }
        if (i >= d->rank && (d->in_spatial[i] != 1 || d->out_spatial[i] != 1 || d->kernel[i] != 1 ||
                             d->stride[i] != 1 || d->dilation[i] != 1 || d->pad_lo[i] != 0)) {
            set_error("axis %d is beyond rank %d and must be the identity (1,1,1,1,1,0)", i, d->rank);
            return QK_ERR_INVALID_ARG;
        }
        si *= d->in_spatial[i]; so *= d->out_spatial[i]; taps *= d->kernel[i];
    }
    // kernels index rows and elements with 32-bit integers
    if (si * d->batch * 4LL * d->cq > INT_MAX || so * d->batch * 4LL * d->fq > INT_MAX ||
        taps * d->cq * 4LL * d->fq > INT_MAX) {
        set_error("tensor with >= 2^31 elements: split the batch"); return QK_ERR_UNSUPPORTED;
    }
    if (d->layout != QK_CH_LAST && d->layout != QK_CH_FIRST) { set_error("bad layout %d", d->layout); return QK_ERR_INVALID_ARG; }
    if (d->dtype != QK_F32 && d->dtype != QK_BF16 && d->dtype != QK_F16) { set_error("bad dtype %d", d->dtype); return QK_ERR_INVALID_ARG; }
    if (d->activation != QK_ACT_LINEAR && d->activation != QK_ACT_RELU) { set_error("bad activation %d", d->activation); return QK_ERR_INVALID_ARG; }
    if (d->kernel_order != QK_KERNEL_TAPS_MAJOR && d->kernel_order != QK_KERNEL_CHANNEL_MAJOR) { set_error("bad kernel_order %d", d->kernel_order); return QK_ERR_INVALID_ARG; }
    if (d->kernel_order == QK_KERNEL_CHANNEL_MAJOR && (d->dtype == QK_F32 || d->layout != QK_CH_LAST || d->cq % 32 || d->fq % 32)) {
        set_error("kernel_order = QK_KERNEL_CHANNEL_MAJOR is served by the 16-bit matrix-core kernels only (16-bit dtype, channels_last, cq and fq multiples of 32)");
        return QK_ERR_UNSUPPORTED;
    }
    return 0;
}

// a channel-major kernel (qk_conv_desc_t.kernel_order) that the 16-bit kernels declined must not fall through to kernels that read taps-major
int refuse_ch_major(const qk_conv_desc_t *d, const char *what)
{
    if (d->kernel_order != QK_KERNEL_CHANNEL_MAJOR) return 0;
    set_error("%s: kernel_order = QK_KERNEL_CHANNEL_MAJOR, but the 16-bit matrix-core path did not take this shape (workspace, alignment or a diagnostic switch)", what);
    return QK_ERR_UNSUPPORTED;
}

struct Strides { long long sn, ss[3], sc; long long flat_ss; };

// element strides of an activation tensor with `ch` channels and spatial extents sp
Strides act_strides(const int32_t *sp, int ch, int layout)
{
    Strides s;
    const long long S = (long long)sp[0] * sp[1] * sp[2];
    if (layout == QK_CH_LAST) {
        s.sc = 1; s.ss[2] = ch; s.ss[1] = (long long)sp[2] * ch; s.ss[0] = (long long)sp[1] * sp[2] * ch;
        s.sn = S * ch; s.flat_ss = ch;
    } else {
        s.sc = S; s.ss[2] = 1; s.ss[1] = sp[2]; s.ss[0] = (long long)sp[1] * sp[2];
        s.sn = S * ch; s.flat_ss = 1;
    }
    return s;
}

int taps_of(const qk_conv_desc_t *d) { return d->kernel[0] * d->kernel[1] * d->kernel[2]; }

// qk_postop_t -> kernel form; `rank` / `sp` describe the tensor the post-op acts on
int to_postop(const qk_postop_t *q, int rank, const int32_t *sp, PostOp *p)
{
    memset(p, 0, sizeof(*p));
    if (!q) return 0;
    if (!(q->drop_rate >= 0.f && q->drop_rate < 1.f)) { set_error("post-op: drop_rate %g out of range", (double)q->drop_rate); return QK_ERR_INVALID_ARG; }
    if (!q->alpha) {
        // relu + dropout: y = drop(relu(pre)), one output tensor, the backward reads y only (qk_postop.h, kind 2)
        p->kind = 2; p->alpha_sel = -1; p->alpha_len = 0; p->alpha = nullptr;
    } else {
        if (q->alpha_axis < -1 || q->alpha_axis >= (rank > 0 ? rank : 1)) {
            set_error("post-op: alpha_axis %d out of range", q->alpha_axis); return QK_ERR_INVALID_ARG;
        }
        const int want = q->alpha_axis < 0 ? 1 : sp[q->alpha_axis];
        if (q->alpha_len != want) { set_error("post-op: alpha_len %d, expected %d", q->alpha_len, want); return QK_ERR_INVALID_ARG; }
        p->kind = 1; p->alpha_sel = q->alpha_axis; p->alpha_len = q->alpha_len; p->alpha = q->alpha;
    }
    p->drop_scale = 1.f / (1.f - q->drop_rate);
    unsigned thr = (unsigned)(q->drop_rate * 256.f + 0.5f);      // 8 random bits per element: rates are multiples of 1/256
    p->drop_thr = thr > 255u ? 255u : thr;
    p->drop_scale = p->drop_thr ? 256.f / (256.f - (float)p->drop_thr) : 1.f;   // the scale of the rate actually applied
    p->drop_seed = q->drop_seed;
    p->seed_dev = q->drop_seed_dev;
    return 0;
}

// the post-op as its own elementwise pass over a channels_last (batch, sp, ch) tensor
int postop_pass(int dtype, bool backward, const PostOp &p, int batch, const int32_t *sp, int ch, const void *pre,
                const void *dy, void *out, float *dalpha, hipStream_t stream)
{
    const long long rows = (long long)batch * sp[0] * sp[1] * sp[2];
    int key_div = 1, key_mod = 1;
    if (p.alpha_sel >= 0) { for (int i = p.alpha_sel + 1; i < 3; ++i) key_div *= sp[i]; key_mod = sp[p.alpha_sel]; }
    const int vec = dtype == QK_F32 ? 4 : 8;
    if (ch % vec != 0 || !aligned(pre, 16) || !aligned(out, 16) || (dy && !aligned(dy, 16))) {
        set_error("post-op: channels must be a multiple of %d and buffers 16-byte aligned", vec); return QK_ERR_UNSUPPORTED;
    }
    if (backward && p.alpha_len > 256) { set_error("post-op: alpha_len %d > 256", p.alpha_len); return QK_ERR_UNSUPPORTED; }
    if (rows * ch >= (1ll << 32)) { set_error("post-op: tensor has >= 2^32 elements"); return QK_ERR_UNSUPPORTED; }
    return launch_postop(dtype, backward, pre, dy, out, dalpha, p, rows, ch, key_div, key_mod, stream);
}

size_t w_floats(const qk_conv_desc_t *d) { return (size_t)taps_of(d) * d->cq * 4 * d->fq; }

size_t dy_bytes(const qk_conv_desc_t *d)
{
    const size_t e = (size_t)d->batch * d->out_spatial[0] * d->out_spatial[1] * d->out_spatial[2] * 4 * d->fq;
    return (e * elem_bytes(d->dtype) + 255) / 256 * 256;
}

// ---- channels_first 16-bit buffers on the matrix cores ----------------------------------------------------------
// The 16-bit MFMA kernels want the reduction axis (channels) contiguous.  A QK_CH_FIRST descriptor with 16-bit
// activations is therefore served by re-laying its operands out to channels_last in the caller's workspace
// (k_relayout16: HBM-bound, 64 x 64 tiles through LDS), running the channels_last kernels -- post-ops, chain flags,
// accumulation included -- and re-laying the result back.  Costs one extra read + write of every activation operand
// (~0.15 ms per 367 MB tensor) instead of the 10x slower general fp32 kernels that served these descriptors before.
// (The engine's own layers keep channels_first tensors physically channels-last and never come here; this is for a
// binder that holds true NCHW buffers.)
bool cf16_ok(const qk_conv_desc_t *d)
{
    if (d->layout != QK_CH_FIRST || d->dtype == QK_F32 || (debug_flags() & kDbgNoMfma16)) return false;
    if (d->cq % 32 || d->fq % 32 || taps_of(d) > 32) return false;
    for (int i = 0; i < 3; ++i) if (d->stride[i] != 1) return false;
    return true;
}
size_t align256(size_t n) { return (n + 255) / 256 * 256; }
size_t x_bytes16(const qk_conv_desc_t *d) { return align256((size_t)d->batch * d->in_spatial[0] * d->in_spatial[1] * d->in_spatial[2] * 4 * d->cq * 2); }
size_t y_bytes16(const qk_conv_desc_t *d) { return align256((size_t)d->batch * d->out_spatial[0] * d->out_spatial[1] * d->out_spatial[2] * 4 * d->fq * 2); }
int in_positions(const qk_conv_desc_t *d) { return d->in_spatial[0] * d->in_spatial[1] * d->in_spatial[2]; }
int out_positions(const qk_conv_desc_t *d) { return d->out_spatial[0] * d->out_spatial[1] * d->out_spatial[2]; }
qk_conv_desc_t as_ch_last(const qk_conv_desc_t *d) { qk_conv_desc_t c = *d; c.layout = QK_CH_LAST; return c; }
// x-shaped / y-shaped tensor: channels_first -> channels_last and back
int x_to_last(const qk_conv_desc_t *d, const void *src, void *dst, hipStream_t s) { return launch_relayout16(src, dst, d->batch, 4 * d->cq, in_positions(d), s); }
int x_to_first(const qk_conv_desc_t *d, const void *src, void *dst, hipStream_t s) { return launch_relayout16(src, dst, d->batch, in_positions(d), 4 * d->cq, s); }
int y_to_last(const qk_conv_desc_t *d, const void *src, void *dst, hipStream_t s) { return launch_relayout16(src, dst, d->batch, 4 * d->fq, out_positions(d), s); }
int y_to_first(const qk_conv_desc_t *d, const void *src, void *dst, hipStream_t s) { return launch_relayout16(src, dst, d->batch, out_positions(d), 4 * d->fq, s); }

void prep_geom(const qk_conv_desc_t *d, bool bwd, GemmGeom *gp);

size_t ws_bytes_impl(const qk_conv_desc_t *d, int op)
{
    if (cf16_ok(d)) {
        // [what the channels_last call needs][re-laid-out operands]: fwd x, y (+ pre of a post-op); bwd-data dy, y, dx,
        // dx mask; bwd-weight x, dy, y; fused bwd x, dy, y, dx (+ x_pre of a post-op) -- and never less than its parts
        const qk_conv_desc_t c = as_ch_last(d);
        const size_t X = x_bytes16(d), Y = y_bytes16(d);
        const size_t fwd = align256(ws_bytes_impl(&c, QK_OP_FWD)) + X + 2 * Y;
        const size_t bd = align256(ws_bytes_impl(&c, QK_OP_BWD_DATA)) + 2 * Y + 2 * X;
        const size_t bw = align256(ws_bytes_impl(&c, QK_OP_BWD_WEIGHT)) + X + 2 * Y;
        const size_t bb = align256(ws_bytes_impl(&c, QK_OP_BWD)) + 2 * X + 2 * Y;
        switch (op) {
        case QK_OP_FWD: return fwd;
        case QK_OP_BWD_DATA: return bd;
        case QK_OP_BWD_WEIGHT: return bw;
        case QK_OP_BWD: { size_t m = bb; if (bd > m) m = bd; if (bw > m) m = bw; return m; }
        }
        return 0;
    }
    // 16-bit fast path (fwd / bwd-data): 16-bit re-laid-out copy of the compact kernel (+ zero line);
    // the fp32-MFMA kernels read the compact kernel in place and need nothing
    // fused backward additionally: the relu-masked copy of dy that bwd-weight writes for bwd-data
    size_t n = 0;
    const bool bwd_data = op == QK_OP_BWD_DATA || op == QK_OP_BWD;
    if (d->dtype != QK_F32 && (op == QK_OP_FWD || bwd_data)) {
        // channel counts that are multiples of 16: zero-padded to the kernels' 32-channel granule (other counts never reach the
        // 16-bit kernels; their figure is what it always was)
        const bool on16 = d->cq % 16 == 0 && d->fq % 16 == 0;
        n += (on16 ? (size_t)taps_of(d) * pad32(d->cq) * 4 * pad32(d->fq) : w_floats(d)) * 2 + 256;
        if (on16 && d->layout == QK_CH_LAST) {        // 16 / 32-channel layers: the fragment layout of k_hconv16_small in a second region (go16)
            GemmGeom sg, sbg;
            Small16 sm;
            prep_geom(d, bwd_data, &sg);
            if (d->kernel_order != QK_KERNEL_CHANNEL_MAJOR && small16_shape(sg, &sbg, &sm)) n += small16_region_bytes(sm);
        }
    }
    if (op == QK_OP_BWD && d->activation == QK_ACT_RELU) n = (n + 255) / 256 * 256 + dy_bytes(d);
    return n;
}

// The geometry fields small16_shape / band_geom look at, filled as conv_fwd_impl / conv_bwd_data_impl fill them (channels_last)
void prep_geom(const qk_conv_desc_t *d, bool bwd, GemmGeom *gp)
{
    GemmGeom &g = *gp;
    memset(&g, 0, sizeof(g));
    const Strides xs = act_strides(d->in_spatial, 4 * d->cq, d->layout);
    const Strides ys = act_strides(d->out_spatial, 4 * d->fq, d->layout);
    g.batch = d->batch;
    g.taps = taps_of(d);
    if (!bwd) {
        g.M = d->batch * d->out_spatial[0] * d->out_spatial[1] * d->out_spatial[2];
        g.Q = d->cq; g.J = d->fq;
        for (int i = 0; i < 3; ++i) {
            g.osp[i] = d->out_spatial[i]; g.isp[i] = d->in_spatial[i]; g.ks[i] = d->kernel[i];
            g.pa[i] = d->stride[i]; g.pb[i] = d->dilation[i]; g.pc[i] = -d->pad_lo[i]; g.pd[i] = 1;
            g.in_ss[i] = xs.ss[i];
        }
        g.in_sn = xs.sn; g.in_sc = xs.sc; g.out_sn = ys.sn; g.out_ss = ys.flat_ss; g.out_sc = ys.sc;
    } else {
        g.M = d->batch * d->in_spatial[0] * d->in_spatial[1] * d->in_spatial[2];
        g.Q = d->fq; g.J = d->cq;
        for (int i = 0; i < 3; ++i) {
            g.osp[i] = d->in_spatial[i]; g.isp[i] = d->out_spatial[i]; g.ks[i] = d->kernel[i];
            g.pa[i] = 1; g.pb[i] = -d->dilation[i]; g.pc[i] = d->pad_lo[i]; g.pd[i] = d->stride[i];
            g.in_ss[i] = ys.ss[i];
        }
        g.in_sn = ys.sn; g.in_sc = ys.sc; g.out_sn = xs.sn; g.out_ss = xs.flat_ss; g.out_sc = xs.sc;
    }
}

int conv_fwd_impl(const qk_conv_desc_t *d, const void *x, const float *w, const float *bias, void *y,
                  void *ws, size_t wsb, hipStream_t stream, const PostOp *post = nullptr, void *pre = nullptr)
{
    if (!x || !w || !y) { set_error("x/w/y must not be NULL"); return QK_ERR_INVALID_ARG; }
    if (d->has_bias && !bias) { set_error("has_bias set but bias is NULL"); return QK_ERR_INVALID_ARG; }
    ProfScope prof_scope(QK_OP_FWD, d, stream);
    if (cf16_ok(d)) {
        const qk_conv_desc_t c = as_ch_last(d);
        const size_t base = align256(ws_bytes_impl(&c, QK_OP_FWD)), X = x_bytes16(d), Y = y_bytes16(d);
        const bool two = post && post->kind == 1;
        if (ws && aligned(ws, 16) && wsb >= base + X + Y + (two ? Y : 0)) {
            char *p = static_cast<char *>(ws) + base;
            void *xt = p, *yt = p + X, *pt = two ? p + X + Y : nullptr;
            if (int rc = x_to_last(d, x, xt, stream)) return rc;
            if (int rc = conv_fwd_impl(&c, xt, w, bias, yt, ws, base, stream, post, pt)) return rc;
            if (int rc = y_to_first(d, yt, y, stream)) return rc;
            return two ? y_to_first(d, pt, pre, stream) : 0;
        }
    }
    GemmGeom g;
    memset(&g, 0, sizeof(g));
    const Strides xs = act_strides(d->in_spatial, 4 * d->cq, d->layout);
    const Strides ys = act_strides(d->out_spatial, 4 * d->fq, d->layout);
    g.batch = d->batch;
    g.M = d->batch * d->out_spatial[0] * d->out_spatial[1] * d->out_spatial[2];
    g.Q = d->cq; g.J = d->fq; g.taps = taps_of(d);
    for (int i = 0; i < 3; ++i) {
        g.osp[i] = d->out_spatial[i]; g.isp[i] = d->in_spatial[i]; g.ks[i] = d->kernel[i];
        g.pa[i] = d->stride[i]; g.pb[i] = d->dilation[i]; g.pc[i] = -d->pad_lo[i]; g.pd[i] = 1;
        g.in_ss[i] = xs.ss[i];
    }
    g.in_sn = xs.sn; g.in_sc = xs.sc;
    g.out_sn = ys.sn; g.out_ss = ys.flat_ss; g.out_sc = ys.sc;
    g.sign_tbl = d->conj ? kSignConj : kSignConv;
    g.relu = d->activation == QK_ACT_RELU;
    g.has_bias = d->has_bias ? 1 : 0;
    g.has_mask = 0;
    g.w_prepped = d->ws_has_kernel ? 1 : 0;
    g.w_ch_major = d->kernel_order == QK_KERNEL_CHANNEL_MAJOR;
    const bool with_post = post && post->kind;
    if (d->dtype != QK_F32) {
        if (with_post && d->layout == QK_CH_LAST && aligned(pre, 16) && aligned(y, 16) &&
            (long long)g.M * 4 * d->fq < (1ll << 32)) {                 // the dropout hash indexes elements with 32 bits
            g.post = *post; g.post_fwd = 1; g.pre_out = post->kind == 1 ? pre : nullptr;
        }
        const int r = try_hgemm_16(d->dtype, x, nullptr, w, bias, y, g, false, ws, wsb, stream);
        if (r != 0) return r < 0 ? r : 0;
        g.post.kind = 0; g.post_fwd = 0; g.pre_out = nullptr;
    }
    if (int rc = refuse_ch_major(d, "forward")) return rc;
    const bool vec = d->layout == QK_CH_LAST && d->cq % 4 == 0 && d->fq % 4 == 0 &&
                     vec_aligned(x, d->dtype) && aligned(w, 16);
    note_path(QK_PATH_FP32_MFMA);
    if (!with_post) return launch_hgemm(d->dtype, x, nullptr, w, bias, y, g, vec, stream);
    // general path: the convolution writes pre (kind 2: into y, in place), the post-op runs as its own pass
    if (d->layout != QK_CH_LAST) { set_error("post-op needs channels_last buffers"); return QK_ERR_UNSUPPORTED; }
    void *lin = post->kind == 1 ? pre : y;
    if (int rc = launch_hgemm(d->dtype, x, nullptr, w, bias, lin, g, vec, stream)) return rc;
    return postop_pass(d->dtype, false, *post, d->batch, d->out_spatial, 4 * d->fq, lin, nullptr, y, nullptr, stream);
}

int conv_bwd_data_impl(const qk_conv_desc_t *d, const void *dy, const void *y, const float *w, void *dx,
                       void *ws, size_t wsb, hipStream_t stream, const void *dx_mask = nullptr,
                       const PostOp *post = nullptr, float *dalpha = nullptr)
{
    // post != NULL: dx_mask holds the PRE-activation of x (x = post(x_pre); kind 2: x itself); dx returns d loss / d x_pre
    const bool with_post = post && post->kind;
    if (!dy || !w || !dx) { set_error("dy/w/dx must not be NULL"); return QK_ERR_INVALID_ARG; }
    const bool mask = d->activation == QK_ACT_RELU;
    if (mask && !y) { set_error("activation is RELU: the forward output y is required"); return QK_ERR_INVALID_ARG; }
    const size_t need = ws_bytes_impl(d, QK_OP_BWD_DATA);
    if (need && (!ws || wsb < need)) {
        set_error("bwd_data needs %zu workspace bytes, got %zu", need, wsb);
        return QK_ERR_WORKSPACE;
    }
    if (need && !aligned(ws, 16)) { set_error("workspace must be 16-byte aligned"); return QK_ERR_WORKSPACE; }
    ProfScope prof_scope(QK_OP_BWD_DATA, d, stream);
    if (cf16_ok(d)) {
        const qk_conv_desc_t c = as_ch_last(d);
        const size_t base = align256(ws_bytes_impl(&c, QK_OP_BWD_DATA)), X = x_bytes16(d), Y = y_bytes16(d);
        char *p = static_cast<char *>(ws) + base;
        void *dyt = p, *yt = mask ? p + Y : nullptr, *dxt = p + 2 * Y, *mt = dx_mask ? p + 2 * Y + X : nullptr;
        if (int rc = y_to_last(d, dy, dyt, stream)) return rc;
        if (mask) if (int rc = y_to_last(d, y, yt, stream)) return rc;
        if (dx_mask) if (int rc = x_to_last(d, dx_mask, mt, stream)) return rc;
        if (int rc = conv_bwd_data_impl(&c, dyt, yt, w, dxt, ws, base, stream, mt, post, dalpha)) return rc;
        return x_to_first(d, dxt, dx, stream);
    }
    GemmGeom g;
    memset(&g, 0, sizeof(g));
    const Strides dys = act_strides(d->out_spatial, 4 * d->fq, d->layout);
    const Strides dxs = act_strides(d->in_spatial, 4 * d->cq, d->layout);
    g.batch = d->batch;
    g.M = d->batch * d->in_spatial[0] * d->in_spatial[1] * d->in_spatial[2];
    g.Q = d->fq; g.J = d->cq; g.taps = taps_of(d);
    for (int i = 0; i < 3; ++i) {
        g.osp[i] = d->in_spatial[i]; g.isp[i] = d->out_spatial[i]; g.ks[i] = d->kernel[i];
        g.pa[i] = 1; g.pb[i] = -d->dilation[i]; g.pc[i] = d->pad_lo[i]; g.pd[i] = d->stride[i];
        g.in_ss[i] = dys.ss[i];
    }
    g.in_sn = dys.sn; g.in_sc = dys.sc;
    g.out_sn = dxs.sn; g.out_ss = dxs.flat_ss; g.out_sc = dxs.sc;
    g.sign_tbl = d->conj ? kSignConv : kSignConj;   // transposed table
    g.relu = 0; g.has_bias = 0; g.has_mask = mask ? 1 : 0;
    g.w_prepped = d->ws_has_kernel ? 1 : 0;
    g.w_ch_major = d->kernel_order == QK_KERNEL_CHANNEL_MAJOR;
    if (d->dtype != QK_F32) {
        // the 16-bit kernels apply an epilogue mask themselves (needs 16-byte aligned rows of dx_mask)
        g.ep_mask = (dx_mask && aligned(dx_mask, 16)) ? dx_mask : nullptr;
        const bool fuse_post = with_post && g.ep_mask && post->alpha_len <= 256 && d->layout == QK_CH_LAST;
        if (fuse_post) { g.post = *post; g.dalpha = dalpha; }
        else if (with_post) g.ep_mask = nullptr;
        const int r = try_hgemm_16(d->dtype, dy, mask ? y : nullptr, w, nullptr, dx, g, true, ws, wsb, stream);
        if (r < 0) return r;
        if (r > 0) {
            if (with_post && !fuse_post)
                return postop_pass(d->dtype, true, *post, d->batch, d->in_spatial, 4 * d->cq, dx_mask, dx, dx, dalpha, stream);
            if (dx_mask && !g.ep_mask) return launch_mask_gt0(d->dtype, dx, dx_mask, (size_t)g.M * 4 * d->cq, stream);
            return 0;
        }
        g.ep_mask = nullptr; g.post.kind = 0; g.dalpha = nullptr;
    }
    if (int rc = refuse_ch_major(d, "backward-data")) return rc;
    // the fp32-MFMA kernel stages the compact kernel in place with the channel/filter roles swapped
    g.w_swapped = 1;
    const bool vec = d->layout == QK_CH_LAST && d->cq % 4 == 0 && d->fq % 4 == 0 && aligned(w, 16) &&
                     vec_aligned(dy, d->dtype) && (!mask || vec_aligned(y, d->dtype));
    note_path(QK_PATH_FP32_MFMA);
    if (int rc = launch_hgemm(d->dtype, dy, mask ? y : nullptr, w, nullptr, dx, g, vec, stream)) return rc;
    if (with_post) {
        if (d->layout != QK_CH_LAST) { set_error("post-op needs channels_last buffers"); return QK_ERR_UNSUPPORTED; }
        return postop_pass(d->dtype, true, *post, d->batch, d->in_spatial, 4 * d->cq, dx_mask, dx, dx, dalpha, stream);
    }
    if (dx_mask) return launch_mask_gt0(d->dtype, dx, dx_mask, (size_t)g.M * 4 * d->cq, stream);
    return 0;
}

int conv_bwd_weight_impl(const qk_conv_desc_t *d, const void *x, const void *dy, const void *y, float *dw,
                         float *dbias, void *dy_masked_out, hipStream_t stream, bool accumulate = false,
                         void *ws = nullptr, size_t wsb = 0)
{
    if (!x || !dy || !dw) { set_error("x/dy/dw must not be NULL"); return QK_ERR_INVALID_ARG; }
    const bool mask = d->activation == QK_ACT_RELU;
    if (mask && !y) { set_error("activation is RELU: the forward output y is required"); return QK_ERR_INVALID_ARG; }
    if (d->has_bias && !dbias) { set_error("has_bias set but dbias is NULL"); return QK_ERR_INVALID_ARG; }
    ProfScope prof_scope(QK_OP_BWD_WEIGHT, d, stream);
    if (cf16_ok(d)) {
        // (ws, wsb): the caller's workspace, used for the re-layout; no masked dy is left behind in this mode
        const qk_conv_desc_t c = as_ch_last(d);
        const size_t X = x_bytes16(d), Y = y_bytes16(d);
        if (ws && aligned(ws, 16) && wsb >= X + Y + (mask ? Y : 0)) {
            char *p = static_cast<char *>(ws);
            void *xt = p, *dyt = p + X, *yt = mask ? p + X + Y : nullptr;
            if (int rc = x_to_last(d, x, xt, stream)) return rc;
            if (int rc = y_to_last(d, dy, dyt, stream)) return rc;
            if (mask) if (int rc = y_to_last(d, y, yt, stream)) return rc;
            return conv_bwd_weight_impl(&c, xt, dyt, yt, dw, dbias, nullptr, stream, accumulate);
        }
        dy_masked_out = nullptr;               // (the workspace may be too small to double as the masked-dy output)
    }
    WgradGeom g;
    memset(&g, 0, sizeof(g));
    const Strides xs = act_strides(d->in_spatial, 4 * d->cq, d->layout);
    const Strides dys = act_strides(d->out_spatial, 4 * d->fq, d->layout);
    g.batch = d->batch;
    g.M = d->batch * d->out_spatial[0] * d->out_spatial[1] * d->out_spatial[2];
    g.Cq = d->cq; g.F = d->fq; g.taps = taps_of(d);
    for (int i = 0; i < 3; ++i) {
        g.osp[i] = d->out_spatial[i]; g.isp[i] = d->in_spatial[i]; g.ks[i] = d->kernel[i];
        g.pa[i] = d->stride[i]; g.pb[i] = d->dilation[i]; g.pc[i] = -d->pad_lo[i];
        g.x_ss[i] = xs.ss[i];
    }
    g.x_sn = xs.sn; g.x_sc = xs.sc;
    g.dy_sn = dys.sn; g.dy_ss = dys.flat_ss; g.dy_sc = dys.sc;
    g.sign_tbl = d->conj ? kSignConj : kSignConv;
    g.has_mask = mask ? 1 : 0;
    g.want_dbias = (d->has_bias && dbias) ? 1 : 0;
    g.dym = mask ? dy_masked_out : nullptr;
    g.w_ch_major = d->kernel_order == QK_KERNEL_CHANNEL_MAJOR;
    // dw / dbias are accumulated atomically: zero them first (one fill when they are adjacent, as in
    // a flat gradient buffer)
    const size_t dwb = w_floats(d) * sizeof(float), dbb = 4 * (size_t)d->fq * sizeof(float);
    const char *dw_end = reinterpret_cast<const char *>(dw) + dwb;
    const char *db_c = reinterpret_cast<const char *>(dbias);
    if (accumulate) {
        // the caller's buffers already hold what this call adds to
    } else if (g.want_dbias && (db_c == dw_end ||
                                (db_c > dw_end && db_c - dw_end < 256 && reinterpret_cast<uintptr_t>(db_c) % 256 == 0))) {
        // adjacent, or dbias on the first 256-byte boundary behind dw (a flat gradient buffer with aligned
        // views): the bytes in between are alignment padding by contract (include/qk.h) -- one fill
        if (hipMemsetAsync(dw, 0, (size_t)(db_c - reinterpret_cast<const char *>(dw)) + dbb, stream) != hipSuccess) { set_error("memset dw failed"); return QK_ERR_LAUNCH; }
    } else {
        if (hipMemsetAsync(dw, 0, dwb, stream) != hipSuccess) { set_error("memset dw failed"); return QK_ERR_LAUNCH; }
        if (g.want_dbias && hipMemsetAsync(dbias, 0, dbb, stream) != hipSuccess) { set_error("memset dbias failed"); return QK_ERR_LAUNCH; }
    }
    if (d->dtype != QK_F32) {
        int r = g.w_ch_major ? 0 : try_wgrad_band_16(d->dtype, x, dy, mask ? y : nullptr, dw, dbias, g, stream);     // (the band kernels write taps-major only)
        if (r == 0) r = try_wgrad_16(d->dtype, x, dy, mask ? y : nullptr, dw, dbias, g, stream);
        if (r != 0) return r < 0 ? r : 0;
    }
    if (int rc = refuse_ch_major(d, "backward-weight")) return rc;
    const bool vec = d->layout == QK_CH_LAST && d->cq % 4 == 0 && d->fq % 4 == 0 &&
                     vec_aligned(x, d->dtype) && vec_aligned(dy, d->dtype) && (!mask || vec_aligned(y, d->dtype));
    note_path(QK_PATH_FP32_MFMA);
    return launch_wgrad(d->dtype, x, dy, mask ? y : nullptr, dw, dbias, g, vec, stream);
}

// Fused backward: bwd-weight first (it reads dy and y once and, for RELU, also writes the masked dy
// into the workspace), then bwd-data on the masked copy with no mask loads of its own.
int conv_bwd_impl(const qk_conv_desc_t *d, const void *x, const void *dy, const void *y, const float *w,
                  void *dx, float *dw, float *dbias, void *ws, size_t wsb, hipStream_t stream, int flags = 0)
{
    if (!dx) { set_error("dx must not be NULL (use qk_*_bwd_weight when d(input) is not needed)"); return QK_ERR_INVALID_ARG; }
    if (flags & ~(QK_BWD_MASK_DX | QK_BWD_DY_PREMASKED | QK_BWD_ACCUMULATE)) { set_error("unknown backward flags 0x%x", flags); return QK_ERR_INVALID_ARG; }
    if (cf16_ok(d)) {
        // one re-layout of x, dy (and y) serves both gradient kernels; dx goes back once
        const qk_conv_desc_t c = as_ch_last(d);
        const size_t base = align256(ws_bytes_impl(&c, QK_OP_BWD)), X = x_bytes16(d), Y = y_bytes16(d);
        const bool need_y = d->activation == QK_ACT_RELU && !(flags & QK_BWD_DY_PREMASKED);
        if (need_y && !y) { set_error("activation is RELU: the forward output y is required"); return QK_ERR_INVALID_ARG; }
        if (!x || !dy) { set_error("x/dy must not be NULL"); return QK_ERR_INVALID_ARG; }
        if (ws && aligned(ws, 16) && wsb >= base + 2 * X + 2 * Y) {
            char *p = static_cast<char *>(ws) + base;
            void *xt = p, *dxt = p + X, *dyt = p + 2 * X, *yt = need_y ? p + 2 * X + Y : nullptr;
            if (int rc = x_to_last(d, x, xt, stream)) return rc;
            if (int rc = y_to_last(d, dy, dyt, stream)) return rc;
            if (need_y) if (int rc = y_to_last(d, y, yt, stream)) return rc;
            if (int rc = conv_bwd_impl(&c, xt, dyt, yt, w, dxt, dw, dbias, ws, base, stream, flags)) return rc;
            return x_to_first(d, dxt, dx, stream);
        }
    }
    const bool acc = (flags & QK_BWD_ACCUMULATE) != 0;
    const void *dx_mask = (flags & QK_BWD_MASK_DX) ? x : nullptr;
    const bool relu = d->activation == QK_ACT_RELU && !(flags & QK_BWD_DY_PREMASKED);
    const size_t bd = ws_bytes_impl(d, QK_OP_BWD_DATA);
    if (!relu) {
        // linear layer, or dy arrives with the relu mask applied: both gradients straight from dy
        if (bd && (!ws || wsb < bd)) { set_error("bwd needs %zu workspace bytes, got %zu", bd, wsb); return QK_ERR_WORKSPACE; }
        qk_conv_desc_t lin = *d;
        lin.activation = QK_ACT_LINEAR;
        if (int rc = conv_bwd_weight_impl(&lin, x, dy, nullptr, dw, dbias, nullptr, stream, acc)) return rc;
        return conv_bwd_data_impl(&lin, dy, nullptr, w, dx, ws, bd, stream, dx_mask);
    }
    const size_t need = ws_bytes_impl(d, QK_OP_BWD);
    if (need && (!ws || wsb < need)) { set_error("bwd needs %zu workspace bytes, got %zu", need, wsb); return QK_ERR_WORKSPACE; }
    void *dym = static_cast<char *>(ws) + (bd + 255) / 256 * 256;
    if ((bd + 255) / 256 * 256 + dy_bytes(d) > wsb + 255) { set_error("workspace too small for the masked dy"); return QK_ERR_WORKSPACE; }
    if (int rc = conv_bwd_weight_impl(d, x, dy, y, dw, dbias, dym, stream, acc)) return rc;
    qk_conv_desc_t lin = *d;
    lin.activation = QK_ACT_LINEAR;
    return conv_bwd_data_impl(&lin, dym, nullptr, w, dx, ws, bd, stream, dx_mask);
}

qk_conv_desc_t dense_as_conv(const qk_dense_desc_t *d)
{
    qk_conv_desc_t c;
    memset(&c, 0, sizeof(c));
    c.rank = 0; c.batch = d->rows; c.cq = d->in_q; c.fq = d->q_units;
    for (int i = 0; i < 3; ++i) {
        c.in_spatial[i] = c.out_spatial[i] = c.kernel[i] = c.stride[i] = c.dilation[i] = 1;
        c.pad_lo[i] = 0;
    }
    c.layout = QK_CH_LAST; c.dtype = d->dtype; c.activation = d->activation;
    c.has_bias = d->has_bias; c.conj = 1;     // dense.py:139-143 is the transposed table
    c.ws_has_kernel = d->ws_has_kernel;
    return c;
}

int check_launch(int rc, const char *what)
{
    if (rc == QK_ERR_LAUNCH) set_error("%s: kernel launch failed: %s", what, hipGetErrorString(hipGetLastError()));
    return rc;
}

}  // namespace
}  // namespace qk

using namespace qk;

extern "C" {

int qk_version(void) { return QK_VERSION; }

const char *qk_last_error(void) { return g_err; }

unsigned qk_set_debug_flags(unsigned flags) { return dbg_word().exchange(flags, std::memory_order_relaxed); }
void qk_set_debug_buffer(void *device_buffer, size_t bytes) { g_dbg_buf.store(static_cast<unsigned long long *>(device_buffer)); g_dbg_bytes.store(bytes); }
unsigned qk_get_debug_flags(void) { return debug_flags(); }
int qk_last_path(void) { return g_path; }

int qk_prof_enable(int on)
{
    Prof &p = prof();
    std::lock_guard<std::mutex> lk(p.mu);
    const int was = p.on.exchange(on ? 1 : 0, std::memory_order_relaxed);
    if (on) {                                   // a new recording: drop the previous one
        for (ProfRec &r : p.recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
        p.recs.clear();
    }
    return was;
}
int qk_prof_count(void)
{
    std::lock_guard<std::mutex> lk(prof().mu);
    return (int)prof().recs.size();
}
int qk_prof_get(int i, qk_prof_rec_t *out)
{
    if (!out) { set_error("qk_prof_get: out is NULL"); return QK_ERR_INVALID_ARG; }
    ProfRec r;
    {
        std::lock_guard<std::mutex> lk(prof().mu);
        if (i < 0 || i >= (int)prof().recs.size()) { set_error("qk_prof_get: record %d of %zu", i, prof().recs.size()); return QK_ERR_INVALID_ARG; }
        r = prof().recs[i];
    }
    float ms = 0.f;
    if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) {
        set_error("qk_prof_get: the events of record %d could not be read", i); return QK_ERR_LAUNCH;
    }
    out->op = r.op; out->dtype = r.dtype; out->path = r.path; out->rows = r.rows; out->n = r.n; out->k = r.k; out->ms = ms;
    return QK_OK;
}

size_t qk_conv_workspace_bytes(const qk_conv_desc_t *desc, int op)
{
    if (validate(desc, false)) return 0;
    return ws_bytes_impl(desc, op);
}

size_t qk_dense_workspace_bytes(const qk_dense_desc_t *desc, int op)
{
    if (!desc) return 0;
    const qk_conv_desc_t c = dense_as_conv(desc);
    if (validate(&c, true)) return 0;
    return ws_bytes_impl(&c, op);
}

int qk_conv_fwd(const qk_conv_desc_t *desc, const void *x, const float *w, const float *bias, void *y,
                void *workspace, size_t workspace_bytes, void *stream)
{
    if (int rc = validate(desc, false)) return rc;
    return check_launch(conv_fwd_impl(desc, x, w, bias, y, workspace, workspace_bytes, (hipStream_t)stream), "qk_conv_fwd");
}

int qk_conv_fwd_post(const qk_conv_desc_t *desc, const qk_postop_t *post, const void *x, const float *w,
                     const float *bias, void *pre, void *y, void *workspace, size_t workspace_bytes, void *stream)
{
    if (int rc = validate(desc, false)) return rc;
    if (!post || (!pre && post->alpha)) { set_error("qk_conv_fwd_post: post must not be NULL, pre only for a relu post-op (alpha == NULL)"); return QK_ERR_INVALID_ARG; }
    if (desc->activation != QK_ACT_LINEAR) { set_error("qk_conv_fwd_post: the convolution must be LINEAR (the post-op is the activation)"); return QK_ERR_INVALID_ARG; }
    PostOp p;
    if (int rc = to_postop(post, desc->rank, desc->out_spatial, &p)) return rc;
    return check_launch(conv_fwd_impl(desc, x, w, bias, y, workspace, workspace_bytes, (hipStream_t)stream, &p, pre), "qk_conv_fwd_post");
}

int qk_conv_bwd_post(const qk_conv_desc_t *desc, const void *x, const void *dy, const float *w, void *dx, float *dw,
                     float *dbias, const qk_postop_t *post_x, const void *x_pre, float *dalpha_x, int32_t flags,
                     void *workspace, size_t workspace_bytes, void *stream)
{
    if (int rc = validate(desc, false)) return rc;
    if (!dx || !post_x) { set_error("qk_conv_bwd_post: dx / post_x must not be NULL"); return QK_ERR_INVALID_ARG; }
    if (post_x->alpha && (!x_pre || !dalpha_x)) { set_error("qk_conv_bwd_post: a PReLU post-op needs x_pre and dalpha_x"); return QK_ERR_INVALID_ARG; }
    if (flags & ~QK_BWD_ACCUMULATE) { set_error("qk_conv_bwd_post: only QK_BWD_ACCUMULATE is a valid flag here (0x%x)", flags); return QK_ERR_INVALID_ARG; }
    if (desc->activation != QK_ACT_LINEAR) { set_error("qk_conv_bwd_post: the layer must be LINEAR"); return QK_ERR_INVALID_ARG; }
    PostOp p;
    if (int rc = to_postop(post_x, desc->rank, desc->in_spatial, &p)) return rc;
    const size_t bd = ws_bytes_impl(desc, QK_OP_BWD_DATA);
    if (bd && (!workspace || workspace_bytes < bd)) { set_error("bwd needs %zu workspace bytes, got %zu", bd, workspace_bytes); return QK_ERR_WORKSPACE; }
    if (int rc = conv_bwd_weight_impl(desc, x, dy, nullptr, dw, dbias, nullptr, (hipStream_t)stream, (flags & QK_BWD_ACCUMULATE) != 0,
                                      workspace, workspace_bytes)) return check_launch(rc, "qk_conv_bwd_post");
    // relu post-op: x = drop(relu(x_pre)) is its own mask (x > 0 <=> x_pre > 0 and kept)
    const void *mask_src = p.kind == 2 ? x : x_pre;
    return check_launch(conv_bwd_data_impl(desc, dy, nullptr, w, dx, workspace, bd, (hipStream_t)stream, mask_src, &p, p.kind == 2 ? nullptr : dalpha_x), "qk_conv_bwd_post");
}

static int postop_entry(const qk_conv_desc_t *t, const qk_postop_t *post, bool backward, const void *pre, const void *dy,
                        void *out, float *dalpha, void *stream)
{
    if (!t || !post || !pre || !out || (backward && (!dy || (!dalpha && post->alpha)))) { set_error("post-op: NULL argument"); return QK_ERR_INVALID_ARG; }
    if (t->rank < 0 || t->rank > 3 || t->batch <= 0 || t->fq <= 0) { set_error("post-op: bad tensor description"); return QK_ERR_INVALID_ARG; }
    int32_t sp[3] = {1, 1, 1};
    for (int i = 0; i < t->rank; ++i) { if (t->out_spatial[i] <= 0) { set_error("post-op: bad extent"); return QK_ERR_INVALID_ARG; } sp[i] = t->out_spatial[i]; }
    PostOp p;
    if (int rc = to_postop(post, t->rank, sp, &p)) return rc;
    return check_launch(postop_pass(t->dtype, backward, p, t->batch, sp, 4 * t->fq, pre, dy, out, dalpha, (hipStream_t)stream), "qk_postop");
}

int qk_postop_fwd(const qk_conv_desc_t *t, const qk_postop_t *post, const void *pre, void *y, void *stream)
{
    return postop_entry(t, post, false, pre, nullptr, y, nullptr, stream);
}

int qk_postop_bwd(const qk_conv_desc_t *t, const qk_postop_t *post, const void *pre, const void *dy, void *dpre,
                  float *dalpha, void *stream)
{
    return postop_entry(t, post, true, pre, dy, dpre, dalpha, stream);
}

int qk_conv_bwd_data(const qk_conv_desc_t *desc, const void *dy, const void *y, const float *w, void *dx,
                     void *workspace, size_t workspace_bytes, void *stream)
{
    if (int rc = validate(desc, false)) return rc;
    return check_launch(conv_bwd_data_impl(desc, dy, y, w, dx, workspace, workspace_bytes, (hipStream_t)stream), "qk_conv_bwd_data");
}

int qk_conv_bwd_weight(const qk_conv_desc_t *desc, const void *x, const void *dy, const void *y, float *dw,
                       float *dbias, void *workspace, size_t workspace_bytes, void *stream)
{
    if (int rc = validate(desc, false)) return rc;
    void *dym = (workspace && workspace_bytes >= dy_bytes(desc) && aligned(workspace, 16)) ? workspace : nullptr;
    return check_launch(conv_bwd_weight_impl(desc, x, dy, y, dw, dbias, dym, (hipStream_t)stream, false, workspace, workspace_bytes), "qk_conv_bwd_weight");
}

int qk_conv_bwd_chain(const qk_conv_desc_t *desc, const void *x, const void *dy, const void *y, const float *w,
                      void *dx, float *dw, float *dbias, int32_t flags, void *workspace, size_t workspace_bytes,
                      void *stream)
{
    if (int rc = validate(desc, false)) return rc;
    if ((flags & QK_BWD_MASK_DX) && !x) { set_error("QK_BWD_MASK_DX needs x"); return QK_ERR_INVALID_ARG; }
    return check_launch(conv_bwd_impl(desc, x, dy, y, w, dx, dw, dbias, workspace, workspace_bytes, (hipStream_t)stream, flags), "qk_conv_bwd_chain");
}

int qk_dense_bwd_chain(const qk_dense_desc_t *desc, const void *x, const void *dy, const void *y, const float *w,
                       void *dx, float *dw, float *dbias, int32_t flags, void *workspace, size_t workspace_bytes,
                       void *stream)
{
    if (!desc) { set_error("descriptor is NULL"); return QK_ERR_INVALID_ARG; }
    const qk_conv_desc_t c = dense_as_conv(desc);
    if (int rc = validate(&c, true)) return rc;
    if ((flags & QK_BWD_MASK_DX) && !x) { set_error("QK_BWD_MASK_DX needs x"); return QK_ERR_INVALID_ARG; }
    return check_launch(conv_bwd_impl(&c, x, dy, y, w, dx, dw, dbias, workspace, workspace_bytes, (hipStream_t)stream, flags), "qk_dense_bwd_chain");
}

int qk_conv_bwd_weight_acc(const qk_conv_desc_t *desc, const void *x, const void *dy, const void *y,
                           float *dw, float *dbias, void *workspace, size_t workspace_bytes, void *stream)
{
    if (int rc = validate(desc, false)) return rc;
    void *dym = (workspace && workspace_bytes >= dy_bytes(desc) && aligned(workspace, 16)) ? workspace : nullptr;
    return check_launch(conv_bwd_weight_impl(desc, x, dy, y, dw, dbias, dym, (hipStream_t)stream, true, workspace, workspace_bytes), "qk_conv_bwd_weight_acc");
}

int qk_dense_bwd_weight_acc(const qk_dense_desc_t *desc, const void *x, const void *dy, const void *y,
                            float *dw, float *dbias, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!desc) { set_error("descriptor is NULL"); return QK_ERR_INVALID_ARG; }
    const qk_conv_desc_t c = dense_as_conv(desc);
    if (int rc = validate(&c, true)) return rc;
    void *dym = (workspace && workspace_bytes >= dy_bytes(&c) && aligned(workspace, 16)) ? workspace : nullptr;
    return check_launch(conv_bwd_weight_impl(&c, x, dy, y, dw, dbias, dym, (hipStream_t)stream, true), "qk_dense_bwd_weight_acc");
}

int qk_conv_bwd(const qk_conv_desc_t *desc, const void *x, const void *dy, const void *y, const float *w,
                void *dx, float *dw, float *dbias, void *workspace, size_t workspace_bytes, void *stream)
{
    if (int rc = validate(desc, false)) return rc;
    return check_launch(conv_bwd_impl(desc, x, dy, y, w, dx, dw, dbias, workspace, workspace_bytes, (hipStream_t)stream), "qk_conv_bwd");
}

int qk_dense_bwd(const qk_dense_desc_t *desc, const void *x, const void *dy, const void *y, const float *w,
                 void *dx, float *dw, float *dbias, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!desc) { set_error("descriptor is NULL"); return QK_ERR_INVALID_ARG; }
    const qk_conv_desc_t c = dense_as_conv(desc);
    if (int rc = validate(&c, true)) return rc;
    return check_launch(conv_bwd_impl(&c, x, dy, y, w, dx, dw, dbias, workspace, workspace_bytes, (hipStream_t)stream), "qk_dense_bwd");
}

int qk_dense_fwd(const qk_dense_desc_t *desc, const void *x, const float *w, const float *bias, void *y,
                 void *workspace, size_t workspace_bytes, void *stream)
{
    if (!desc) { set_error("descriptor is NULL"); return QK_ERR_INVALID_ARG; }
    const qk_conv_desc_t c = dense_as_conv(desc);
    if (int rc = validate(&c, true)) return rc;
    return check_launch(conv_fwd_impl(&c, x, w, bias, y, workspace, workspace_bytes, (hipStream_t)stream), "qk_dense_fwd");
}

int qk_dense_bwd_data(const qk_dense_desc_t *desc, const void *dy, const void *y, const float *w, void *dx,
                      void *workspace, size_t workspace_bytes, void *stream)
{
    if (!desc) { set_error("descriptor is NULL"); return QK_ERR_INVALID_ARG; }
    const qk_conv_desc_t c = dense_as_conv(desc);
    if (int rc = validate(&c, true)) return rc;
    return check_launch(conv_bwd_data_impl(&c, dy, y, w, dx, workspace, workspace_bytes, (hipStream_t)stream), "qk_dense_bwd_data");
}

int qk_dense_bwd_weight(const qk_dense_desc_t *desc, const void *x, const void *dy, const void *y, float *dw,
                        float *dbias, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!desc) { set_error("descriptor is NULL"); return QK_ERR_INVALID_ARG; }
    const qk_conv_desc_t c = dense_as_conv(desc);
    if (int rc = validate(&c, true)) return rc;
    void *dym = (workspace && workspace_bytes >= dy_bytes(&c) && aligned(workspace, 16)) ? workspace : nullptr;
    return check_launch(conv_bwd_weight_impl(&c, x, dy, y, dw, dbias, dym, (hipStream_t)stream), "qk_dense_bwd_weight");
}

static bool conv1_pool_ok(const qk_conv_desc_t *d, int32_t pool, int act = QK_ACT_RELU)
{
    // layout QK_CH_FIRST here describes x ONLY: (N, 4, H, W), the four component planes; the pooled tensor is channels_last
    return d->rank == 2 && (d->dtype == QK_BF16 || d->dtype == QK_F16) && d->cq == 1 &&
           d->kernel[0] == 3 && d->kernel[1] == 5 && d->stride[0] == 1 && d->stride[1] == 1 && d->dilation[0] == 1 &&
           d->dilation[1] == 1 && d->pad_lo[0] == 1 && d->pad_lo[1] == 2 && d->out_spatial[0] == d->in_spatial[0] &&
           d->out_spatial[1] == d->in_spatial[1] && d->activation == act && d->conj == 0 && d->fq % 8 == 0 && pool == 3 &&        // (fq: 16-byte runs of filters per component; blocks of 32, the last one partly used)
           // the kernel's windows are rows [3o, 3o + 2]: TensorFlow's 'same' pooling (window = stride = 3) pads one row
           // on the LOW side when H % 3 == 1 (total pad 2 -> 1 + 1), so those heights are not this kernel's
           d->in_spatial[0] % 3 != 1 &&
           (long long)d->batch * ((d->in_spatial[0] + 2) / 3) * d->in_spatial[1] * 4 * d->fq < INT_MAX;
}

size_t qk_conv_relu_pool_aux_bytes(const qk_conv_desc_t *desc, int32_t pool)
{
    if (validate(desc, false) || !(conv1_pool_ok(desc, pool) || conv1_pool_ok(desc, pool, QK_ACT_LINEAR))) return 0;
    return conv1_pool_argbits_bytes(desc->batch, desc->in_spatial[0], desc->in_spatial[1], desc->fq);
}

int qk_conv_relu_pool_fwd(const qk_conv_desc_t *desc, int32_t pool, const void *x, const float *w, const float *bias,
                          void *pooled, void *aux, void *stream)
{
    if (int rc = validate(desc, false)) return rc;
    if (!conv1_pool_ok(desc, pool)) { set_error("qk_conv_relu_pool: geometry outside the fused first-layer kernel (see include/qk.h)"); return QK_ERR_UNSUPPORTED; }
    if (!x || !w || !pooled || (desc->has_bias && !bias)) { set_error("qk_conv_relu_pool_fwd: NULL argument"); return QK_ERR_INVALID_ARG; }
    if (!aligned(x, 8) || !aligned(pooled, 16) || (aux && !aligned(aux, 16))) { set_error("qk_conv_relu_pool_fwd: alignment"); return QK_ERR_INVALID_ARG; }
    note_path(QK_PATH_MFMA16);
    ProfScope prof_scope(QK_OP_FWD, desc, (hipStream_t)stream);
    return check_launch(launch_conv1_pool(desc->dtype, false, x, w, bias, pooled, aux, nullptr, nullptr, desc->batch, desc->in_spatial[0],
                                          desc->in_spatial[1], desc->fq, desc->has_bias, (hipStream_t)stream, nullptr, 0, nullptr, nullptr,
                                          desc->layout == QK_CH_FIRST), "qk_conv_relu_pool_fwd");
}

int qk_conv_relu_pool_bwd(const qk_conv_desc_t *desc, int32_t pool, const void *x, const void *dpooled, const void *aux,
                          float *dw, float *dbias, int32_t flags, void *stream)
{
    if (int rc = validate(desc, false)) return rc;
    if (flags & ~QK_BWD_ACCUMULATE) { set_error("qk_conv_relu_pool_bwd: only QK_BWD_ACCUMULATE is a valid flag (0x%x)", flags); return QK_ERR_INVALID_ARG; }
    if (!conv1_pool_ok(desc, pool)) { set_error("qk_conv_relu_pool: geometry outside the fused first-layer kernel (see include/qk.h)"); return QK_ERR_UNSUPPORTED; }
    if (!x || !dpooled || !aux || !dw || (desc->has_bias && !dbias)) { set_error("qk_conv_relu_pool_bwd: NULL argument"); return QK_ERR_INVALID_ARG; }
    if (!aligned(x, 8) || !aligned(dpooled, 16) || !aligned(aux, 16)) { set_error("qk_conv_relu_pool_bwd: alignment"); return QK_ERR_INVALID_ARG; }
    hipStream_t st = (hipStream_t)stream;
    note_path(QK_PATH_MFMA16);
    ProfScope prof_scope(QK_OP_BWD_WEIGHT, desc, st);
    if (!(flags & QK_BWD_ACCUMULATE) &&
        (hipMemsetAsync(dw, 0, w_floats(desc) * sizeof(float), st) != hipSuccess ||
         (desc->has_bias && hipMemsetAsync(dbias, 0, 4 * (size_t)desc->fq * sizeof(float), st) != hipSuccess))) {
        set_error("memset dw failed"); return QK_ERR_LAUNCH;
    }
    return check_launch(launch_conv1_pool(desc->dtype, true, x, nullptr, nullptr, dpooled, const_cast<void *>(aux), dw, desc->has_bias ? dbias : nullptr,
                                          desc->batch, desc->in_spatial[0], desc->in_spatial[1], desc->fq, desc->has_bias, st, nullptr, 0, nullptr,
                                          nullptr, desc->layout == QK_CH_FIRST), "qk_conv_relu_pool_bwd");
}

// PReLU form of the fused first layer: slopes per row of the conv output (alpha_axis 0, alpha_len == in_spatial[0] <= 64)
// or one slope (alpha_axis -1); no dropout (the reference has none between the first convolution and its pooling)
static int prelu_pool_post_ok(const qk_conv_desc_t *desc, const qk_postop_t *post)
{
    if (!post || !post->alpha) { set_error("qk_conv_prelu_pool: post / alpha is NULL"); return QK_ERR_INVALID_ARG; }
    if (post->drop_rate != 0.f) { set_error("qk_conv_prelu_pool: dropout is not part of this kernel"); return QK_ERR_UNSUPPORTED; }
    if (!((post->alpha_axis == -1 && post->alpha_len == 1) || (post->alpha_axis == 0 && post->alpha_len == desc->in_spatial[0] && post->alpha_len <= 64))) {
        set_error("qk_conv_prelu_pool: slopes must be one, or one per position of spatial axis 0 (at most 64)"); return QK_ERR_UNSUPPORTED;
    }
    return QK_OK;
}

int qk_conv_prelu_pool_fwd(const qk_conv_desc_t *desc, int32_t pool, const qk_postop_t *post, const void *x, const float *w,
                           const float *bias, void *pooled, void *pre_pooled, void *aux, void *stream)
{
    if (int rc = validate(desc, false)) return rc;
    if (!conv1_pool_ok(desc, pool, QK_ACT_LINEAR)) { set_error("qk_conv_prelu_pool: geometry outside the fused first-layer kernel (see include/qk.h)"); return QK_ERR_UNSUPPORTED; }
    if (int rc = prelu_pool_post_ok(desc, post)) return rc;
    if (!x || !w || !pooled || (desc->has_bias && !bias)) { set_error("qk_conv_prelu_pool_fwd: NULL argument"); return QK_ERR_INVALID_ARG; }
    if (!aligned(x, 8) || !aligned(pooled, 16) || (aux && !aligned(aux, 16)) || (pre_pooled && !aligned(pre_pooled, 16))) { set_error("qk_conv_prelu_pool_fwd: alignment"); return QK_ERR_INVALID_ARG; }
    note_path(QK_PATH_MFMA16);
    ProfScope prof_scope(QK_OP_FWD, desc, (hipStream_t)stream);
    return check_launch(launch_conv1_pool(desc->dtype, false, x, w, bias, pooled, aux, nullptr, nullptr, desc->batch, desc->in_spatial[0],
                                          desc->in_spatial[1], desc->fq, desc->has_bias, (hipStream_t)stream, post->alpha, post->alpha_len,
                                          pre_pooled, nullptr, desc->layout == QK_CH_FIRST), "qk_conv_prelu_pool_fwd");
}

int qk_conv_prelu_pool_bwd(const qk_conv_desc_t *desc, int32_t pool, const qk_postop_t *post, const void *x, const void *dpooled,
                           const void *pre_pooled, const void *aux, float *dw, float *dbias, float *dalpha, int32_t flags, void *stream)
{
    if (int rc = validate(desc, false)) return rc;
    if (flags & ~QK_BWD_ACCUMULATE) { set_error("qk_conv_prelu_pool_bwd: only QK_BWD_ACCUMULATE is a valid flag (0x%x)", flags); return QK_ERR_INVALID_ARG; }
    if (!conv1_pool_ok(desc, pool, QK_ACT_LINEAR)) { set_error("qk_conv_prelu_pool: geometry outside the fused first-layer kernel (see include/qk.h)"); return QK_ERR_UNSUPPORTED; }
    if (int rc = prelu_pool_post_ok(desc, post)) return rc;
    if (!x || !dpooled || !pre_pooled || !aux || !dw || (desc->has_bias && !dbias)) { set_error("qk_conv_prelu_pool_bwd: NULL argument"); return QK_ERR_INVALID_ARG; }
    if (!aligned(x, 8) || !aligned(dpooled, 16) || !aligned(pre_pooled, 16) || !aligned(aux, 16)) { set_error("qk_conv_prelu_pool_bwd: alignment"); return QK_ERR_INVALID_ARG; }
    hipStream_t st = (hipStream_t)stream;
    note_path(QK_PATH_MFMA16);
    ProfScope prof_scope(QK_OP_BWD_WEIGHT, desc, st);
    if (!(flags & QK_BWD_ACCUMULATE) &&
        (hipMemsetAsync(dw, 0, w_floats(desc) * sizeof(float), st) != hipSuccess ||
         (desc->has_bias && hipMemsetAsync(dbias, 0, 4 * (size_t)desc->fq * sizeof(float), st) != hipSuccess))) {
        set_error("memset dw failed"); return QK_ERR_LAUNCH;
    }
    return check_launch(launch_conv1_pool(desc->dtype, true, x, nullptr, nullptr, dpooled, const_cast<void *>(aux), dw, desc->has_bias ? dbias : nullptr,
                                          desc->batch, desc->in_spatial[0], desc->in_spatial[1], desc->fq, desc->has_bias, st, post->alpha,
                                          post->alpha_len, pre_pooled, dalpha, desc->layout == QK_CH_FIRST), "qk_conv_prelu_pool_bwd");
}

int qk_conv_fold_taps(const qk_conv_desc_t *desc, const void *x, void *xcol, int32_t cq2, void *stream)
{
    if (int rc = validate(desc, false)) return rc;
    if (!x || !xcol) { set_error("x/xcol must not be NULL"); return QK_ERR_INVALID_ARG; }
    if (cq2 % 8 != 0 || cq2 < taps_of(desc) * desc->cq) {
        set_error("cq2 = %d must be a multiple of 8 and >= taps*cq = %d", cq2, taps_of(desc) * desc->cq);
        return QK_ERR_INVALID_ARG;
    }
    GemmGeom g;
    memset(&g, 0, sizeof(g));
    const Strides xs = act_strides(desc->in_spatial, 4 * desc->cq, desc->layout);
    g.batch = desc->batch;
    g.M = desc->batch * desc->out_spatial[0] * desc->out_spatial[1] * desc->out_spatial[2];
    g.Q = desc->cq; g.taps = taps_of(desc);
    for (int i = 0; i < 3; ++i) {
        g.osp[i] = desc->out_spatial[i]; g.isp[i] = desc->in_spatial[i]; g.ks[i] = desc->kernel[i];
        g.pa[i] = desc->stride[i]; g.pb[i] = desc->dilation[i]; g.pc[i] = -desc->pad_lo[i]; g.pd[i] = 1;
        g.in_ss[i] = xs.ss[i];
    }
    g.in_sn = xs.sn; g.in_sc = xs.sc;
    if ((long long)g.M * 4 * cq2 > INT_MAX) { set_error("folded tensor has >= 2^31 elements"); return QK_ERR_UNSUPPORTED; }
    return check_launch(launch_fold_taps(desc->dtype, x, xcol, g, cq2, (hipStream_t)stream), "qk_conv_fold_taps");
}

static int pool_geom(const qk_pool_desc_t *d, PoolGeom *g, const void *a, const void *b, const void *c)
{
    if (!d) { set_error("descriptor is NULL"); return QK_ERR_INVALID_ARG; }
    if (!a || !b || !c) { set_error("maxpool: NULL buffer"); return QK_ERR_INVALID_ARG; }
    if (d->batch <= 0 || d->in_h <= 0 || d->in_w <= 0 || d->channels <= 0 || d->win_h <= 0 || d->win_w <= 0 ||
        d->out_h <= 0 || d->out_w <= 0) {
        set_error("maxpool: extents must be positive"); return QK_ERR_INVALID_ARG;
    }
    if (d->dtype != QK_F32 && d->dtype != QK_BF16 && d->dtype != QK_F16) { set_error("maxpool: bad dtype"); return QK_ERR_INVALID_ARG; }
    // windows tile the input from position 0; only the last one may be partial
    if ((long long)(d->out_h - 1) * d->win_h >= d->in_h || (long long)(d->out_w - 1) * d->win_w >= d->in_w) {
        set_error("maxpool: out extents %dx%d do not fit %dx%d with window %dx%d", d->out_h, d->out_w, d->in_h, d->in_w, d->win_h, d->win_w);
        return QK_ERR_INVALID_ARG;
    }
    const int v = d->dtype == QK_F32 ? 4 : 8;
    if (d->channels % v != 0) { set_error("maxpool: channels must be a multiple of %d", v); return QK_ERR_UNSUPPORTED; }
    if (!aligned(a, 16) || !aligned(b, 16) || !aligned(c, 16)) { set_error("maxpool: buffers must be 16-byte aligned"); return QK_ERR_INVALID_ARG; }
    if ((long long)d->batch * d->in_h * d->in_w * d->channels > INT_MAX) { set_error("maxpool: tensor has >= 2^31 elements"); return QK_ERR_UNSUPPORTED; }
    g->batch = d->batch; g->ih = d->in_h; g->iw = d->in_w; g->C = d->channels;
    g->wh = d->win_h; g->ww = d->win_w; g->oh = d->out_h; g->ow = d->out_w;
    return 0;
}

int qk_maxpool2d_fwd(const qk_pool_desc_t *desc, const void *x, void *y, void *stream)
{
    PoolGeom g;
    if (int rc = pool_geom(desc, &g, x, y, y)) return rc;
    return check_launch(launch_maxpool(desc->dtype, false, x, nullptr, y, g, (hipStream_t)stream), "qk_maxpool2d_fwd");
}

int qk_maxpool2d_bwd(const qk_pool_desc_t *desc, const void *x, const void *dy, void *dx, void *stream)
{
    PoolGeom g;
    if (int rc = pool_geom(desc, &g, x, dy, dx)) return rc;
    // 'valid' pooling leaves trailing rows / columns outside every window: their gradient is zero
    if ((long long)g.oh * g.wh < g.ih || (long long)g.ow * g.ww < g.iw) {
        const size_t es = desc->dtype == QK_F32 ? 4 : 2;
        if (hipMemsetAsync(dx, 0, (size_t)g.batch * g.ih * g.iw * g.C * es, (hipStream_t)stream) != hipSuccess) {
            set_error("maxpool: memset failed"); return QK_ERR_LAUNCH;
        }
    }
    return check_launch(launch_maxpool(desc->dtype, true, x, dy, dx, g, (hipStream_t)stream), "qk_maxpool2d_bwd");
}

size_t qk_ctc_workspace_bytes(int32_t batch, int32_t frames, int32_t max_label_len)
{
    if (batch <= 0 || frames <= 0 || max_label_len < 0) return 0;
    return ctc_workspace_bytes(batch, frames, max_label_len);
}

int qk_ctc_batch_cost(int32_t dtype, int32_t batch, int32_t frames, int32_t classes, const void *y_pred, const int32_t *labels,
                      int32_t max_label_len, const int32_t *input_length, const int32_t *label_length, float *cost, void *dy_pred,
                      void *workspace, size_t workspace_bytes, void *stream)
{
    if (batch <= 0 || frames <= 0 || classes < 2 || max_label_len < 0) { set_error("ctc: bad extents (%d, %d, %d, %d)", batch, frames, classes, max_label_len); return QK_ERR_INVALID_ARG; }
    if (!y_pred || !input_length || !label_length || !cost || (max_label_len > 0 && !labels)) { set_error("ctc: NULL argument"); return QK_ERR_INVALID_ARG; }
    const size_t need = ctc_workspace_bytes(batch, frames, max_label_len);
    if (!workspace || workspace_bytes < need || !aligned(workspace, 4)) { set_error("ctc needs %zu workspace bytes, got %zu", need, workspace_bytes); return QK_ERR_WORKSPACE; }
    if ((long long)batch * frames * classes > INT_MAX) { set_error("ctc: tensor with >= 2^31 elements"); return QK_ERR_UNSUPPORTED; }
    const int rc = launch_ctc(dtype, batch, frames, classes, y_pred, labels, max_label_len, input_length, label_length, cost, dy_pred,
                              static_cast<float *>(workspace), (hipStream_t)stream);
    if (rc == QK_ERR_UNSUPPORTED) set_error("ctc: more than 127 labels, more than 256 classes or too many frames for one workgroup's LDS");
    return check_launch(rc, "qk_ctc_batch_cost");
}

int qk_softmax_rows_fwd(int32_t dtype, int64_t rows, int32_t cols, const float *logits, const float *bias, void *y, void *stream)
{
    if (!logits || !y || rows < 0 || cols < 1 || cols > 64 || dtype < QK_F32 || dtype > QK_F16) { set_error("qk_softmax_rows_fwd: bad argument (1 <= cols <= 64)"); return QK_ERR_INVALID_ARG; }
    if (rows == 0) return QK_OK;
    return check_launch(launch_softmax_rows(dtype, false, logits, bias, y, nullptr, rows, cols, (hipStream_t)stream), "qk_softmax_rows_fwd");
}

int qk_softmax_rows_bwd(int32_t dtype, int64_t rows, int32_t cols, const void *y, const void *dy, void *dlogits, float *dbias, void *stream)
{
    if (!y || !dy || !dlogits || rows < 0 || cols < 1 || cols > 64 || dtype < QK_F32 || dtype > QK_F16) { set_error("qk_softmax_rows_bwd: bad argument (1 <= cols <= 64)"); return QK_ERR_INVALID_ARG; }
    if (rows == 0) return QK_OK;
    return check_launch(launch_softmax_rows(dtype, true, y, dy, dlogits, dbias, rows, cols, (hipStream_t)stream), "qk_softmax_rows_bwd");
}

int qk_dense_softmax_supported(int32_t dtype, int64_t rows, int32_t in_dim, int32_t units)
{
    return dense_softmax_supported(dtype, rows, in_dim, units) ? 1 : 0;
}

int qk_dense_softmax_fwd(int32_t dtype, int64_t rows, int32_t in_dim, int32_t units, const void *x, const float *kernel, const float *bias,
                         void *y, void *stream)
{
    if (!x || !kernel || !y || rows < 0) { set_error("qk_dense_softmax_fwd: NULL buffer or negative row count"); return QK_ERR_INVALID_ARG; }
    if (!dense_softmax_supported(dtype, rows, in_dim, units)) { set_error("qk_dense_softmax_fwd: in_dim %d / units %d / dtype %d outside the kernel (in_dim 64, 128, 256; even units <= 64; 16-bit)", in_dim, units, dtype); return QK_ERR_UNSUPPORTED; }
    if (!aligned(x, 16) || !aligned(y, 4) || !aligned(kernel, 4)) { set_error("qk_dense_softmax_fwd: x must be 16-byte aligned, y 4-byte"); return QK_ERR_INVALID_ARG; }
    if (rows == 0) return QK_OK;
    return check_launch(launch_dense_softmax_fwd(dtype, rows, in_dim, units, x, kernel, bias, y, (hipStream_t)stream), "qk_dense_softmax_fwd");
}

size_t qk_dense_softmax_bwd_workspace_bytes(int32_t dtype, int64_t rows, int32_t in_dim, int32_t units)
{
    return dense_softmax_bwd_workspace_bytes(dtype, rows, in_dim, units);
}

int qk_dense_softmax_bwd(int32_t dtype, int64_t rows, int32_t in_dim, int32_t units, const void *x, const float *kernel, const void *y,
                         const void *dy, void *dx, float *dkernel, float *dbias, const float *dy_scale_dev, float dy_scale,
                         void *workspace, size_t workspace_bytes, void *stream)
{
    if (!x || !kernel || !y || !dy || !dx || rows < 0) { set_error("qk_dense_softmax_bwd: NULL buffer or negative row count"); return QK_ERR_INVALID_ARG; }
    if (!dense_softmax_supported(dtype, rows, in_dim, units)) { set_error("qk_dense_softmax_bwd: in_dim %d / units %d / dtype %d outside the kernel (in_dim 64, 128, 256; even units <= 64; 16-bit)", in_dim, units, dtype); return QK_ERR_UNSUPPORTED; }
    if (!aligned(x, 16) || !aligned(dx, 16) || !aligned(y, 4) || !aligned(dy, 4)) { set_error("qk_dense_softmax_bwd: x / dx must be 16-byte aligned, y / dy 4-byte"); return QK_ERR_INVALID_ARG; }
    if (rows == 0) return QK_OK;
    if (dkernel || dbias) {
        const size_t need = dense_softmax_bwd_workspace_bytes(dtype, rows, in_dim, units);
        if (!workspace || workspace_bytes < need || !aligned(workspace, 16)) { set_error("qk_dense_softmax_bwd needs %zu workspace bytes (16-byte aligned), got %zu", need, workspace_bytes); return QK_ERR_WORKSPACE; }
    }
    return check_launch(launch_dense_softmax_bwd(dtype, rows, in_dim, units, x, kernel, y, dy, dx, dkernel, dbias, dy_scale_dev, dy_scale, static_cast<float *>(workspace), (hipStream_t)stream), "qk_dense_softmax_bwd");
}

int qk_weighted_sum(int32_t dtype, int64_t n, const void *a, const float *w, float *out, void *stream)
{
    if (!a || !w || !out || n < 0 || dtype < QK_F32 || dtype > QK_F16) { set_error("qk_weighted_sum: bad argument"); return QK_ERR_INVALID_ARG; }
    if (n == 0) return QK_OK;
    return check_launch(launch_weighted_sum(dtype, a, w, out, n, (hipStream_t)stream), "qk_weighted_sum");
}

int qk_conv_prep_kernels(int32_t n, const qk_conv_desc_t *const *descs, const int32_t *ops, const float *const *w,
                         void *const *workspaces, void *stream)
{
    if (n < 0 || (n > 0 && (!descs || !ops || !w || !workspaces))) { set_error("qk_conv_prep_kernels: NULL argument"); return QK_ERR_INVALID_ARG; }
    for (int dt = QK_BF16; dt <= QK_F16; ++dt) {
        PrepJobs jobs;
        int m = 0;
        for (int i = 0; i <= n; ++i) {
            if (i < n) {
                const qk_conv_desc_t *d = descs[i];
                if (!d || !w[i] || !workspaces[i]) { set_error("qk_conv_prep_kernels: job %d has a NULL member", i); return QK_ERR_INVALID_ARG; }
                if (d->dtype != dt) continue;
                if (ops[i] != QK_OP_FWD && ops[i] != QK_OP_BWD_DATA && ops[i] != QK_OP_BWD) { set_error("qk_conv_prep_kernels: job %d: bad op %d", i, ops[i]); return QK_ERR_INVALID_ARG; }
                if (!aligned(workspaces[i], 16)) { set_error("qk_conv_prep_kernels: job %d: workspace not 16-byte aligned", i); return QK_ERR_INVALID_ARG; }
                if (d->cq % 16 || d->fq % 16) continue;       // outside the matrix-core path: its calls run the fp32-MFMA kernels, which never read the workspace
                const bool bwd = ops[i] != QK_OP_FWD;
                PrepJob &j = jobs.j[m++];
                j.w = w[i]; j.wq = workspaces[i]; j.taps = taps_of(d); j.cq = d->cq; j.fq = d->fq;
                j.transposed = bwd ? 1 : 0;
                j.small = 0; j.kin = 0; j.n_ot = 0;
                j.ch_major = d->kernel_order == QK_KERNEL_CHANNEL_MAJOR;
                j.neg_ijk = bwd ? (d->conj ? 1 : 0) : (d->conj ? 0 : 1);       // the sign table go16 folds into the kernel
                {   // 16 / 32-channel layers: a second job writes the fragment layout of k_hconv16_small into the region behind the band
                    // layout (the shape alone decides that the region exists, exactly as go16 and qk_conv_workspace_bytes see it)
                    GemmGeom sg, sbg;
                    Small16 sm;
                    prep_geom(d, bwd, &sg);
                    if (d->layout == QK_CH_LAST && !j.ch_major && small16_shape(sg, &sbg, &sm)) {
                        PrepJob &k = jobs.j[m++];
                        k = j;
                        k.wq = static_cast<char *>(workspaces[i]) + (size_t)j.taps * pad32(d->cq) * 4 * pad32(d->fq) * 2 + 256;
                        k.small = 1; k.kin = sm.kin; k.n_ot = sm.n_ot;
                    }
                }
            }
            if (m >= 31 || (i == n && m > 0)) {
                if (int rc = launch_prep_w16_batch(dt, jobs, m, (hipStream_t)stream)) return check_launch(rc, "qk_conv_prep_kernels");
                m = 0;
            }
        }
    }
    for (int i = 0; i < n; ++i)
        if (descs[i]->dtype != QK_BF16 && descs[i]->dtype != QK_F16) { set_error("qk_conv_prep_kernels: job %d is not a 16-bit descriptor", i); return QK_ERR_INVALID_ARG; }
    return QK_OK;
}

int qk_adam_step(float *param, const float *grad, float *m, float *v, size_t n, float lr, float beta1,
                 float beta2, float eps, int32_t step, float grad_scale, void *stream)
{
    if (!param || !grad || !m || !v) { set_error("adam: NULL buffer"); return QK_ERR_INVALID_ARG; }
    if (step < 1) { set_error("adam: step must be >= 1"); return QK_ERR_INVALID_ARG; }
    return check_launch(launch_adam(param, const_cast<float *>(grad), m, v, nullptr, n, lr, beta1, beta2, eps, step, grad_scale, false, (hipStream_t)stream), "qk_adam_step");
}

int qk_adam_step_zero_grad(float *param, float *grad, float *m, float *v, size_t n, float lr, float beta1,
                           float beta2, float eps, int32_t step, float grad_scale, void *stream)
{
    if (!param || !grad || !m || !v) { set_error("adam: NULL buffer"); return QK_ERR_INVALID_ARG; }
    if (step < 1) { set_error("adam: step must be >= 1"); return QK_ERR_INVALID_ARG; }
    return check_launch(launch_adam(param, grad, m, v, nullptr, n, lr, beta1, beta2, eps, step, grad_scale, true, (hipStream_t)stream), "qk_adam_step_zero_grad");
}

int qk_adam_step_l2(float *param, float *grad, float *m, float *v, const float *decay, size_t n, float lr, float beta1,
                    float beta2, float eps, int32_t step, float grad_scale, int32_t zero_grad, void *stream)
{
    if (!param || !grad || !m || !v) { set_error("adam: NULL buffer"); return QK_ERR_INVALID_ARG; }
    if (step < 1) { set_error("adam: step must be >= 1"); return QK_ERR_INVALID_ARG; }
    return check_launch(launch_adam(param, grad, m, v, decay, n, lr, beta1, beta2, eps, step, grad_scale, zero_grad != 0, (hipStream_t)stream), "qk_adam_step_l2");
}

int qk_adam_step_dev(float *param, float *grad, float *m, float *v, const float *decay, size_t n, float lr, float beta1,
                     float beta2, float eps, int32_t *step_dev, float grad_scale, int32_t zero_grad, void *stream)
{
    if (!param || !grad || !m || !v || !step_dev) { set_error("adam: NULL buffer"); return QK_ERR_INVALID_ARG; }
    return check_launch(launch_adam(param, grad, m, v, decay, n, lr, beta1, beta2, eps, 1, grad_scale, zero_grad != 0, (hipStream_t)stream, step_dev), "qk_adam_step_dev");
}

}  // extern "C"
