/*
 * qk.h -- C-ABI of the MI355X (gfx950) quaternion-layer engine (libqk_hip.so).
 *
 * This is the drop-in boundary for the Hamilton-product hot path of
 * Orkis-Research/Quaternion-CNN-for-E2E-ASR.  The reference has no FFI of its own: the path
 * is ~60 lines of Keras-backend calls inside two Layer.call bodies.  Each entry point below
 * names the reference code it replaces (paths relative to the reference checkout):
 *
 *   qk_conv_fwd          QuaternionConv.call      complexnn/conv.py:288-345
 *                        (slices :294-307, signed concat :327-331, K.conv{1,2,3}d :334,
 *                         K.bias_add :336-341, activation :342-343) -- all fused in one launch,
 *                         the 4x-expanded kernel is never materialised.
 *   qk_conv_bwd_data     TF autodiff of the above w.r.t. the layer input
 *   qk_conv_bwd_weight   TF autodiff of the above w.r.t. `kernel` (conv.py:175-181) and
 *                        `bias` (conv.py:270-278): the 16 expanded blocks are folded back onto
 *                        the 4 compact parts inside the kernel.
 *   qk_dense_fwd         QuaternionDense.call     complexnn/dense.py:126-164
 *                        (signed concat :139-143 == TRANSPOSED table => conj(W) (x) x,
 *                         K.dot :149, bias/activation :159-162)
 *   qk_dense_bwd_data / qk_dense_bwd_weight       TF autodiff of dense.py:126-164
 *   qk_conv_bwd / qk_dense_bwd                    the same gradients in one fused call
 *   qk_conv_fold_taps    (no reference counterpart) im2col-style fold that turns a few-channel conv into
 *                        a 1x1 quaternion conv, so the first layer also runs on the MFMA kernels
 *   qk_maxpool2d_fwd/bwd MaxPooling2D between the TIMIT convolutions (models/interspeech_model.py:99-103),
 *                        channels_last, non-overlapping windows: HBM-bound helper
 *   qk_adam_step         the Keras Adam update the reference trains with
 *                        (working_example.py:106, keras.optimizers.Adam defaults) applied to a
 *                        flat fp32 parameter buffer -- used by the data-parallel step.
 *
 * Conventions (identical to the reference):
 *   - quaternion components r,i,j,k are four CONTIGUOUS channel blocks: input channel
 *     a*Cq+c, output channel b*F+f, bias index b*F+f;
 *   - compact kernel layout (*kernel_size, Cq, 4F) with last axis p*F+f (conv.py:165,
 *     init.py:91); dense kernel (in_q, 4*q_units) (dense.py:98, init.py:153);
 *   - activations are channels_last (N, *spatial, 4C) or channels_first (N, 4C, *spatial),
 *     dense activations are (M, 4C) row-major;
 *   - cross-correlation (no kernel flip), explicit low-side padding per axis (the caller
 *     resolves 'valid' / 'same' / 'causal' with TensorFlow's rule and passes pad_lo and the
 *     output extents).
 *
 * Ownership / threading: the caller owns EVERY buffer including the workspace; the library
 * never allocates or frees device memory and never keeps a pointer past the call.  All work is
 * enqueued on the hipStream_t passed in (no host synchronisation); the functions are
 * re-entrant.  Kernel / bias / their gradients are always float32 (Keras floatx); activations
 * (x, y, dy, dx) are float32, bfloat16 or float16 with float32 accumulation.
 *
 * Errors: every function returns 0 on success or a negative qk_status_t; qk_last_error()
 * returns a thread-local message.  Nothing throws across this boundary.
 */
#ifndef QK_H_
#define QK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QK_VERSION 103 /* major*10000 + minor*100 + patch */

typedef enum {
    QK_OK = 0,
    QK_ERR_INVALID_ARG = -1,   /* bad descriptor / null pointer                      */
    QK_ERR_UNSUPPORTED = -2,   /* valid request this build has no kernel for         */
    QK_ERR_WORKSPACE = -3,     /* workspace missing or too small                     */
    QK_ERR_LAUNCH = -4         /* HIP runtime reported an error on launch            */
} qk_status_t;

typedef enum { QK_F32 = 0, QK_BF16 = 1, QK_F16 = 2 } qk_dtype_t;
typedef enum { QK_CH_LAST = 0, QK_CH_FIRST = 1 } qk_layout_t;
typedef enum { QK_ACT_LINEAR = 0, QK_ACT_RELU = 1 } qk_act_t;
typedef enum { QK_OP_FWD = 0, QK_OP_BWD_DATA = 1, QK_OP_BWD_WEIGHT = 2, QK_OP_BWD = 3 } qk_op_t;

/* One quaternion convolution call (QuaternionConv.__init__/build state, conv.py:93-286). */
typedef struct {
    int32_t rank;            /* 1, 2 or 3 spatial axes                                        */
    int32_t batch;           /* N                                                             */
    int32_t in_spatial[3];   /* input extents; trailing unused axes must be 1                 */
    int32_t out_spatial[3];  /* conv_utils.conv_output_length per axis (conv.py:347-372)      */
    int32_t cq;              /* quaternion input channels  = input channels / 4 (conv.py:164) */
    int32_t fq;              /* `filters`: quaternion filters, output channels = 4*fq         */
    int32_t kernel[3];       /* kernel_size                                                   */
    int32_t stride[3];       /* strides                                                       */
    int32_t dilation[3];     /* dilation_rate                                                 */
    int32_t pad_lo[3];       /* zeros in front of each axis ('same': total/2, 'causal': d(k-1)) */
    int32_t layout;          /* qk_layout_t of x / y / dy / dx                                */
    int32_t dtype;           /* qk_dtype_t  of x / y / dy / dx                                */
    int32_t activation;      /* qk_act_t fused into fwd; bwd masks dy with (y > 0) for RELU   */
    int32_t has_bias;        /* use_bias                                                      */
    int32_t conj;            /* 0: W (x) x  (conv.py:327-331);  1: conj(W) (x) x (dense table) */
    int32_t ws_has_kernel;   /* 16-bit path: nonzero = the start of `workspace` ALREADY holds the re-laid-out 16-bit
                              * kernel an earlier call with the same descriptor, the same operation class (forward |
                              * backward-data / fused backward) and UNCHANGED weights left there: the call skips that
                              * step (one small launch per call; 26 per TIMIT training step).  0 = always safe.    */
    int32_t kernel_order;    /* memory order of the compact kernel w / dw.  0 (QK_KERNEL_TAPS_MAJOR): (*kernel_size, cq, 4 fq), the
                              * layer's own layout (conv.py:165).  1 (QK_KERNEL_CHANNEL_MAJOR, round 6): (cq, *kernel_size, 4 fq) -- the
                              * weight of a QuaternionDense applied to the flattened (channel, position) axes of a channels-first feature
                              * map, read IN PLACE as the kernel of the equivalent 'valid' convolution (the TIMIT model's first
                              * TimeDistributed dense layer, models/interspeech_model.py:140-149: row cq * F + f of the dense weight is tap f,
                              * channel cq): no permuted copy of the parameter per step, and the kernel gradient is accumulated straight
                              * into the parameter's own layout.  Served by the 16-bit matrix-core kernels only (channels_last,
                              * cq % 32 == 0, fq % 32 == 0, not a band / small-channel shape for backward-weight): QK_ERR_UNSUPPORTED
                              * otherwise -- never a silent mis-read. */
} qk_conv_desc_t;
#define QK_KERNEL_TAPS_MAJOR 0
#define QK_KERNEL_CHANNEL_MAJOR 1

/* One quaternion dense call (QuaternionDense state, dense.py:58-124). */
typedef struct {
    int32_t rows;            /* M: batch rows (B, or B*T under TimeDistributed)               */
    int32_t in_q;            /* input_dim = in_features / 4          (dense.py:96)            */
    int32_t q_units;         /* units / 4                            (dense.py:75)            */
    int32_t dtype;           /* qk_dtype_t of x / y / dy / dx                                 */
    int32_t activation;      /* qk_act_t                                                      */
    int32_t has_bias;
    int32_t ws_has_kernel;   /* as in qk_conv_desc_t                                          */
} qk_dense_desc_t;

/* Library / diagnostics ------------------------------------------------------------------ */
int qk_version(void);
const char *qk_last_error(void);

/* Diagnostic switches (tests, profiling): a process-wide bit mask, read with one relaxed atomic load on
 * every launch path.  Its initial value comes from the environment, read once (QK_NO_MFMA16, QK_NO_BAND16,
 * QK_NO_BAND32, QK_WGRAD16_ONE_TAP set => the bit; QK_ABLATE=<n> => bits 8..15).  Every combination computes the
 * same values (up to the rounding of the path selected); none is needed in production.
 *   QK_DBG_NO_MFMA16        16-bit activations run on the general fp32-MFMA kernels (exact fp32 products)
 *   QK_DBG_NO_BAND16/32     no band variants of the forward / backward-data kernels
 *   QK_DBG_WGRAD16_ONE_TAP  16-bit backward-weight: one tap per block for 32-channel layers
 *   QK_DBG_BAND16_8WAVES    16-bit band kernels in their 8-wave form (env QK_BAND16_8WAVES)
 *   QK_DBG_NO_WGRAD_BAND    16-bit backward-weight without the band kernel (env QK_NO_WGRAD_BAND)
 *   QK_DBG_NO_POINT16       16-bit one-tap-per-row shapes (1 x 1, head backward-data) on the implicit-GEMM kernel
 *                           instead of the streaming point-form kernel (env QK_NO_POINT16)
 *   bits 8..15              kernel ablation for profiling (skip the MFMA loop / epilogue / atomics): WRONG RESULTS
 * qk_set_debug_flags returns the previous mask. */
#define QK_DBG_NO_MFMA16 1u
#define QK_DBG_NO_BAND16 2u
#define QK_DBG_NO_BAND32 4u
#define QK_DBG_WGRAD16_ONE_TAP 8u
#define QK_DBG_NO_WGRAD_BAND 32u   /* 16-bit backward-weight: one block per tap (k_wgrad16) instead of the band kernel */
#define QK_DBG_NO_POINT16 64u
#define QK_DBG_CTC_TWO_SWEEPS 128u /* qk_ctc_batch_cost: the round-3 form (alpha then beta, gradient inside the backward sweep; env QK_CTC_TWO_SWEEPS) */
#define QK_DBG_DETERMINISTIC 0x10000u /* bit-reproducible gradients (env QK_DETERMINISTIC): every backward-weight kernel runs
                                         * ONE split of the reduction per gradient tile and one owner per bias column, so
                                         * each element of dw / dbias receives exactly one (atomic) addition -- no
                                         * order-dependent float sums.  Several times slower (the reduction over positions is
                                         * no longer spread over the CUs): for debugging and for repeatability tests. */
#define QK_DBG_WGRAD_BAND_V1 0x20000u /* 16-bit backward-weight band kernel in its round-2..4 form (register staging, two tile buffers; env QK_WGRAD_BAND_V1) -- A/B */
#define QK_DBG_NO_SMALL16 0x40000u /* 16-bit layers with 16 / 32 channels per component on the zero-padded band kernels instead of k_hconv16_small
                                     * (env QK_NO_SMALL16; A/B).  Safe to toggle at run time: such a layer's workspace carries both 16-bit
                                     * kernel layouts in regions of their own (round 6) */
/* Graph-level A/B switches (round 6: they were environment variables read by the Python host at call time).  The library itself
 * does not act on them -- they live in this mask so that every A/B switch of the engine has ONE home, ONE initialisation from the
 * environment (same names: QK_NO_CONV_CHAIN ...) and ONE run-time API; the host-side model code (models/interspeech_model.py,
 * layers.py) asks qk_get_debug_flags().  Each one selects the unfused composition of the same arithmetic. */
#define QK_DBG_NO_CONV_CHAIN 0x100000u     /* body convolutions as separate autograd nodes instead of qk_conv_bwd_chain */
#define QK_DBG_NO_FUSED_PRELU 0x200000u    /* PReLU (+ Dropout) as qk_postop_* passes instead of the conv epilogues */
#define QK_DBG_NO_FUSED_DROPOUT 0x400000u  /* relu + Dropout as separate passes instead of the one-output post-op */
#define QK_DBG_NO_FUSED_CTC 0x800000u      /* K.ctc_batch_cost through torch's ctc_loss instead of qk_ctc_batch_cost */
#define QK_DBG_NO_FUSED_FIRST 0x1000000u   /* first layer + frequency pooling as conv, activation and pooling passes instead of qk_conv_*_pool_* */
#define QK_DBG_NO_DENSE_IN_CHAIN 0x2000000u /* the TimeDistributed quaternion-dense layers outside the backward chain */
#define QK_DBG_NO_FUSED_SOFTMAX 0x4000000u /* Dense(62, softmax) through torch ops instead of the fused output-layer kernels */
#define QK_DBG_BAND16_8WAVES 16u   /* 16-bit band kernels: 8-wave workgroups (one per CU) instead of 4-wave (two per CU) */
unsigned qk_set_debug_flags(unsigned flags);
/* Profiling only: a device buffer (64 bytes per workgroup, 65536 workgroups) into which the 16-bit band kernels drop shader-clock time stamps of their
 * phases (start, prologue done, K loop done, end) and the CU they ran on -- probe builds only (tools/probe/build_variant.sh); NULL switches it off. */
void qk_set_debug_buffer(void *device_buffer, size_t bytes);
unsigned qk_get_debug_flags(void);

/* Which kernel family served the most recent qk_* compute call of the calling thread (thread-local, like
 * qk_last_error): lets a caller see when a shape fell off the 16-bit matrix-core fast path. */
typedef enum { QK_PATH_NONE = 0, QK_PATH_MFMA16 = 1, QK_PATH_MFMA16_BAND = 2, QK_PATH_FP32_MFMA = 3,
               QK_PATH_MFMA16_POINT = 4, QK_PATH_MFMA16_SMALL = 5 /* k_hconv16_small: 16 / 32 channels per component */ } qk_path_t;
int qk_last_path(void);

/* Per-call timing for benchmarks.  While enabled, every forward / backward-data / backward-weight call (including the
 * two halves of qk_*_bwd and the fused first-layer calls) is bracketed by a pair of HIP events on the caller's stream;
 * the records stay in a process-wide list until the next qk_prof_enable(1).  `ms` covers everything the call launched
 * (kernel re-layout, memsets, the GEMM kernel); rows x n x k is the GEMM view of the layer (rows = output positions,
 * n = 4 fq, k = taps * 4 cq), so 2 * rows * n * k / ms is the call's algorithmic rate.  qk_prof_get waits for the
 * record's events (host synchronisation: call it after the region of interest).  Off by default; costs one atomic load
 * per call when off. */
typedef struct {
    int32_t op;        /* QK_OP_FWD, QK_OP_BWD_DATA or QK_OP_BWD_WEIGHT */
    int32_t dtype;
    int32_t path;      /* qk_path_t that served it */
    int64_t rows;
    int32_t n, k;
    float ms;
} qk_prof_rec_t;
int qk_prof_enable(int on);          /* returns the previous state; enabling clears the list */
int qk_prof_count(void);
int qk_prof_get(int index, qk_prof_rec_t *out);

/* Bytes of caller-owned device workspace `op` needs for this descriptor (0 = none). */
size_t qk_conv_workspace_bytes(const qk_conv_desc_t *desc, int op /* qk_op_t */);
size_t qk_dense_workspace_bytes(const qk_dense_desc_t *desc, int op /* qk_op_t */);

/* Convolution ----------------------------------------------------------------------------
 * x      [N, *in_spatial, 4*cq]  (or channels_first)        desc->dtype
 * w      [*kernel, cq, 4*fq]                                float32, compact r|i|j|k
 * bias   [4*fq] or NULL                                     float32
 * y      [N, *out_spatial, 4*fq] (or channels_first)        desc->dtype
 * stream hipStream_t (passed as void* so this header needs no HIP include)
 */
int qk_conv_fwd(const qk_conv_desc_t *desc, const void *x, const float *w, const float *bias,
                void *y, void *workspace, size_t workspace_bytes, void *stream);

/* dx = d loss / d x.  `y` (the forward output) is read only when activation == RELU. */
int qk_conv_bwd_data(const qk_conv_desc_t *desc, const void *dy, const void *y, const float *w,
                     void *dx, void *workspace, size_t workspace_bytes, void *stream);

/* dw [*kernel, cq, 4*fq] and dbias [4*fq] (NULL when !has_bias) are OVERWRITTEN.
 * Layout contract: when dbias starts on the first 256-byte boundary behind the end of dw (less than 256
 * bytes away), the bytes between the two are alignment padding OWNED BY THIS CALL and are zeroed with
 * them (one fill instead of two).  Any other placement touches dw and dbias only.
 * Optional: with activation == RELU and a 16-byte-aligned workspace of at least |dy| bytes (rounded
 * up to 256), the masked gradient dy * (y > 0) is left at the start of the workspace in dy's layout;
 * qk_conv_bwd_data may then be called on it with activation = LINEAR (no y reads).  This is how a
 * data-parallel step starts its gradient all-reduce before bwd-data. */
int qk_conv_bwd_weight(const qk_conv_desc_t *desc, const void *x, const void *dy, const void *y,
                       float *dw, float *dbias, void *workspace, size_t workspace_bytes,
                       void *stream);

/* Accumulating form: dw += ..., dbias += ... (nothing is zeroed first).  For gradient accumulation over
 * micro-batches, and for training loops whose optimiser step leaves the gradient buffer zeroed
 * (qk_adam_step_zero_grad): the 5 us fill in front of every backward-weight then disappears. */
int qk_conv_bwd_weight_acc(const qk_conv_desc_t *desc, const void *x, const void *dy, const void *y,
                           float *dw, float *dbias, void *workspace, size_t workspace_bytes,
                           void *stream);
int qk_dense_bwd_weight_acc(const qk_dense_desc_t *desc, const void *x, const void *dy, const void *y,
                            float *dw, float *dbias, void *workspace, size_t workspace_bytes,
                            void *stream);

/* Fused backward: all three gradients in one call (what TF autodiff of conv.py:288-345 yields).
 * bwd-weight runs first and, for RELU, leaves the masked dy in the workspace so that bwd-data reads
 * one tensor instead of two.  dx must not be NULL.  Workspace: qk_*_workspace_bytes(desc, QK_OP_BWD). */
int qk_conv_bwd(const qk_conv_desc_t *desc, const void *x, const void *dy, const void *y, const float *w,
                void *dx, float *dw, float *dbias, void *workspace, size_t workspace_bytes, void *stream);

/* Fused backward inside a chain of relu layers.  The relu derivative of layer L, dy * (y_L > 0), costs
 * backward-weight a second input stream (every tap block re-reads y_L); the cheap place for it is the
 * epilogue of layer L+1's backward-data, whose input x IS y_L.  Flags:
 *   QK_BWD_MASK_DX      dx is returned as dx * (x > 0): the gradient w.r.t. the pre-activation of the layer
 *                       that produced x (valid when x is a relu output);
 *   QK_BWD_DY_PREMASKED dy already carries this layer's relu mask (its consumer was called with
 *                       QK_BWD_MASK_DX): the mask is not applied again and y is not read (may be NULL).
 * With flags == 0 this is qk_conv_bwd / qk_dense_bwd.  Applying a mask twice is harmless (idempotent);
 * omitting QK_BWD_DY_PREMASKED is therefore always safe, setting it on an unmasked dy is not. */
#define QK_BWD_MASK_DX 1
#define QK_BWD_DY_PREMASKED 2
#define QK_BWD_ACCUMULATE 4   /* dw / dbias are ADDED to (as qk_*_bwd_weight_acc): gradients land in a caller-zeroed buffer */
int qk_conv_bwd_chain(const qk_conv_desc_t *desc, const void *x, const void *dy, const void *y, const float *w,
                      void *dx, float *dw, float *dbias, int32_t flags, void *workspace, size_t workspace_bytes,
                      void *stream);
int qk_dense_bwd_chain(const qk_dense_desc_t *desc, const void *x, const void *dy, const void *y, const float *w,
                       void *dx, float *dw, float *dbias, int32_t flags, void *workspace, size_t workspace_bytes,
                       void *stream);

/* PReLU + Dropout behind a layer, as the reference's TIMIT model applies them after every convolution
 * (models/interspeech_model.py:99-101,117-121: `PReLU(shared_axes=[1,0])`, `Dropout(d.dropout)`):
 *     y = drop(prelu(pre)),  prelu(v) = v > 0 ? v : alpha * v,  drop(v) = keep ? v / (1 - rate) : 0.
 * alpha: float32 on the device, ONE scalar (alpha_axis = -1) or one slope per position along spatial axis
 * `alpha_axis` (0 .. rank-1) of the tensor -- what Keras builds for shared_axes=[1,0] on a channels_first (C, F, T)
 * activation is (1, F, 1): alpha_axis = 0, alpha_len = F.  The dropout mask is never stored: keep(e) is a hash of
 * (drop_seed, flat element index of y in its channels_last buffer), identical in forward and backward; pass a new
 * seed every step.  Each element gets 8 random bits: the rate applied is round(drop_rate * 256) / 256 (the kept
 * elements are scaled by the reciprocal of THAT keep probability).  drop_rate = 0 disables dropout.
 *
 * alpha == NULL selects the relu form,  y = drop(relu(pre)),  the model's `aact == 'none'` setting (relu layers with
 * `Dropout(d.dropout)` behind every body convolution, interspeech_model.py:117-121,131-137).  There the slope is 0 by
 * definition, so ONE tensor is written (y; `pre` arguments may be NULL) and the backward needs nothing but y:
 * d pre = dy / (1 - rate) where y > 0  (y > 0 <=> pre > 0 and kept) -- no mask regeneration, no slope gradient
 * (alpha_axis / alpha_len are ignored, dalpha arguments may be NULL). */
typedef struct {
    int32_t alpha_axis;      /* -1: scalar; 0..2: spatial axis of the activation that indexes alpha       */
    int32_t alpha_len;       /* 1 for a scalar, else the extent of that axis                                */
    const float *alpha;      /* device pointer, float32; NULL: relu (+ dropout), see above                  */
    float drop_rate;         /* in [0, 1)                                                                   */
    uint32_t drop_seed;
    const uint32_t *drop_seed_dev; /* optional (NULL: off): a DEVICE counter; the kernels hash with drop_seed + 0x9E3779B1 * (*drop_seed_dev).
                                    * With the training step's number there (qk_adam_step_dev keeps it) every launch argument is the same
                                    * from step to step -- a captured graph replays and still draws new masks; forward and backward of
                                    * one step agree because the counter moves only between steps */
} qk_postop_t;

/* y = post(W (x) x + b): the convolution must be LINEAR (desc->activation); `pre` receives W (x) x + b (same
 * shape / dtype as y; needed by the backward of the post-op; NULL for the relu form), `y` the activated / dropped
 * tensor. */
int qk_conv_fwd_post(const qk_conv_desc_t *desc, const qk_postop_t *post, const void *x, const float *w,
                     const float *bias, void *pre, void *y, void *workspace, size_t workspace_bytes, void *stream);

/* Fused backward of a LINEAR layer (dy = d loss / d pre of THIS layer) whose input x = post_x(x_pre) was produced by
 * a post-op: dw, dbias as qk_conv_bwd; dx receives d loss / d x_pre (the post-op's derivative is applied in the
 * epilogue of backward-data), dalpha_x[alpha_len] (float32) is ACCUMULATED into (zero it once per step).  For the
 * relu form (post_x->alpha == NULL) x_pre and dalpha_x are not used (may be NULL): x is its own mask.
 * flags: 0 or QK_BWD_ACCUMULATE (dw / dbias are added to, as qk_conv_bwd_weight_acc). */
int qk_conv_bwd_post(const qk_conv_desc_t *desc, const void *x, const void *dy, const float *w, void *dx, float *dw,
                     float *dbias, const qk_postop_t *post_x, const void *x_pre, float *dalpha_x, int32_t flags,
                     void *workspace, size_t workspace_bytes, void *stream);

/* The post-op on its own (any dtype / shape, HBM-bound): `t` describes the channels_last activation
 * (batch, spatial[0..rank-1], channels) -- only batch, rank, out_spatial, fq (channels = 4 * fq) and dtype of a
 * qk_conv_desc_t are read.  bwd: dpre = d loss / d pre, dalpha accumulated.  Relu form (alpha == NULL): the
 * backward takes the forward OUTPUT y in place of `pre`, dalpha may be NULL. */
int qk_postop_fwd(const qk_conv_desc_t *t, const qk_postop_t *post, const void *pre, void *y, void *stream);
int qk_postop_bwd(const qk_conv_desc_t *t, const qk_postop_t *post, const void *pre, const void *dy, void *dpre,
                  float *dalpha, void *stream);

/* The first layer of the TIMIT model as ONE kernel per direction (models/interspeech_model.py:97-103):
 *     QuaternionConv2D(F, (3,5), padding='same', relu) on ONE quaternion channel -> MaxPooling2D over the first
 *     spatial axis with window = stride = `pool`, 'same' (partial last window).
 * The layer is HBM-bound; fused, the forward reads x (N, H, W, 4) and writes only the pooled tensor
 * (N, ceil(H / pool), W, 4F) plus `aux` (opaque to the caller, qk_conv_relu_pool_aux_bytes: 3 bits per pooled element --
 * one-hot "window row 0 / 1 / 2 held the maximum and relu let it through", all clear = nothing flows back -- in the
 * kernels' register order), the backward reads x, the pooled gradient and aux and returns dw / dbias (overwritten) -- the
 * 537 MB pre-pool activation of the B = 256 model never exists.  Supported: rank 2, QK_CH_LAST, bf16 / fp16, cq == 1, kernel
 * (3,5), unit stride / dilation, pad_lo (1,2), activation RELU, conj 0, fq % 8 == 0, pool == 3, H % 3 != 1 (the
 * kernel's windows are rows [3o, 3o + 2]; TensorFlow's 'same' rule pads one row on the LOW side when H % 3 == 1, so
 * for those heights the windows would start at row -1: not this kernel's -- 41 bins are fine); anything else returns
 * QK_ERR_UNSUPPORTED (qk_conv_relu_pool_aux_bytes: 0) and the caller runs qk_conv_fwd + qk_maxpool2d_* instead.
 * `aux` may be NULL in the forward (inference).
 * desc->layout describes x ONLY here: QK_CH_LAST = (N, H, W, 4); QK_CH_FIRST = (N, 4, H, W), the four r / i / j / k
 * component PLANES of the one quaternion channel exactly as the reference feeds its model (Input(shape=(4, 41, None)),
 * interspeech_model.py:81) -- each plane row is read with coalesced loads and the planes are interleaved while the
 * patch is staged in LDS.  The pooled tensor (and its gradient) is always channels_last: this layer is where a
 * channels_first model's data enters the engine's channels-last domain, so no separate re-layout pass exists.
 * bwd flags: 0 (dw / dbias overwritten) or QK_BWD_ACCUMULATE (added to). */
size_t qk_conv_relu_pool_aux_bytes(const qk_conv_desc_t *desc, int32_t pool);
int qk_conv_relu_pool_fwd(const qk_conv_desc_t *desc, int32_t pool, const void *x, const float *w, const float *bias,
                          void *pooled, void *aux, void *stream);
int qk_conv_relu_pool_bwd(const qk_conv_desc_t *desc, int32_t pool, const void *x, const void *dpooled, const void *aux,
                          float *dw, float *dbias, int32_t flags, void *stream);

/* The same layer in its PReLU form (the reference's aact == 'prelu'): LINEAR convolution (desc->activation), PReLU with
 * one slope per row of the conv output (post->alpha_axis 0, alpha_len == in_spatial[0] <= 64: Keras shared_axes=[1,0])
 * or one slope (alpha_axis -1), then the pooling; post->drop_rate must be 0.  `pre_pooled` (same shape / dtype as
 * `pooled`) receives the pre-activation of each window's arg-max -- the backward needs it for the derivative and the
 * slope gradient `dalpha[alpha_len]` (float32, ACCUMULATED into); aux as above (qk_conv_relu_pool_aux_bytes accepts
 * the LINEAR descriptor as well).  dw / dbias are overwritten. */
int qk_conv_prelu_pool_fwd(const qk_conv_desc_t *desc, int32_t pool, const qk_postop_t *post, const void *x, const float *w,
                           const float *bias, void *pooled, void *pre_pooled, void *aux, void *stream);
int qk_conv_prelu_pool_bwd(const qk_conv_desc_t *desc, int32_t pool, const qk_postop_t *post, const void *x, const void *dpooled,
                           const void *pre_pooled, const void *aux, float *dw, float *dbias, float *dalpha, int32_t flags,
                           void *stream);

/* Tap folding for layers with very few input channels (the first TIMIT layer has cq = 1: K = 4*taps).
 *   xcol[m, a*cq2 + t*cq + c] = x[pos(m, t), a*cq + c]      (0 in the padding and for t*cq + c >= taps*cq)
 * xcol is channels_last (N, *out_spatial, 4*cq2), cq2 a multiple of 8 with cq2 >= taps*cq.  The layer
 * then IS a 1x1 quaternion convolution on xcol with the kernel reshaped to (taps*cq -> cq2, 4*fq): the
 * Hamilton structure acts per channel, so it survives the fold.  HBM-bound gather. */
int qk_conv_fold_taps(const qk_conv_desc_t *desc, const void *x, void *xcol, int32_t cq2, void *stream);

/* Dense ---------------------------------------------------------------------------------
 * x [rows, 4*in_q], w [in_q, 4*q_units] float32, bias [4*q_units], y [rows, 4*q_units]
 */
int qk_dense_fwd(const qk_dense_desc_t *desc, const void *x, const float *w, const float *bias,
                 void *y, void *workspace, size_t workspace_bytes, void *stream);
int qk_dense_bwd_data(const qk_dense_desc_t *desc, const void *dy, const void *y, const float *w,
                      void *dx, void *workspace, size_t workspace_bytes, void *stream);
int qk_dense_bwd_weight(const qk_dense_desc_t *desc, const void *x, const void *dy, const void *y,
                        float *dw, float *dbias, void *workspace, size_t workspace_bytes,
                        void *stream);

int qk_dense_bwd(const qk_dense_desc_t *desc, const void *x, const void *dy, const void *y, const float *w,
                 void *dx, float *dw, float *dbias, void *workspace, size_t workspace_bytes, void *stream);

/* Max pooling of a channels_last activation (N, H, W, C) -- the layer between the first and second
 * TIMIT convolutions (MaxPooling2D((1,3), padding='same'), models/interspeech_model.py:99-103), which
 * on a channels_first quaternion tensor pools the frequency axis.  Not part of the Hamilton product;
 * it lives here because the engine keeps channels_first tensors physically channels_last and the
 * framework kernels for that layout run at a third of HBM speed.  Windows must not overlap
 * (stride == window) and TensorFlow's padding must fall on the high side only ('valid', or 'same'
 * with no low-side padding): a partial last window simply ignores the missing elements.
 * Ties go to the first maximum in (h, w) window order, NaN propagates (torch / TF semantics).
 * bwd recomputes the arg-max from x (no index tensor) and writes EVERY element of dx. */
typedef struct {
    int32_t batch, in_h, in_w, channels;   /* x: (batch, in_h, in_w, channels)                   */
    int32_t win_h, win_w;                  /* window == stride                                    */
    int32_t out_h, out_w;                  /* ceil(in/win) for 'same', floor(in/win) for 'valid'  */
    int32_t dtype;                         /* qk_dtype_t                                          */
} qk_pool_desc_t;
int qk_maxpool2d_fwd(const qk_pool_desc_t *desc, const void *x, void *y, void *stream);
int qk_maxpool2d_bwd(const qk_pool_desc_t *desc, const void *x, const void *dy, void *dx, void *stream);

/* Optimiser step on a flat fp32 buffer (Keras Adam: working_example.py:106).
 *   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g^2
 *   p -= lr * sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)          (Keras 2.x form)
 * `grad_scale` multiplies g first (1/world_size after a sum all-reduce). */
int qk_adam_step(float *param, const float *grad, float *m, float *v, size_t n, float lr,
                 float beta1, float beta2, float eps, int32_t step, float grad_scale,
                 void *stream);

/* qk_adam_step that also writes zeros over `grad` once it has been consumed, so the next backward can
 * accumulate into it (qk_*_bwd_weight_acc) without a separate fill. */
int qk_adam_step_zero_grad(float *param, float *grad, float *m, float *v, size_t n, float lr,
                           float beta1, float beta2, float eps, int32_t step, float grad_scale,
                           void *stream);

/* K.ctc_batch_cost(y_true, y_pred, input_length, label_length) -- what the reference's TIMIT model outputs
 * (models/interspeech_model.py:37-39,178) -- and its gradient, as one launch: y_pred (batch, frames, classes) are the
 * model's softmax outputs (blank = classes - 1), labels (batch, max_label_len) int32 padded, input_length / label_length
 * (batch) int32.  Keras 2.x / TensorFlow semantics: log(y_pred + 1e-7) taken as LOGITS (normalised again per frame),
 * ctc_merge_repeated = True, frames >= input_length ignored.  cost (batch) float32 = -log p(labels | y_pred);
 * dy_pred (same dtype / shape as y_pred, or NULL) = d cost[b] / d y_pred[b].  One workgroup per sample; limits:
 * Degenerate samples: an infeasible one (more labels, counting a blank between repeats, than frames -- TensorFlow raises
 * for it) gets cost = +inf and a ZERO gradient row; input_length == 0 gives cost 0 for an empty label sequence, +inf
 * otherwise; labels outside [0, classes - 2] are clamped (TensorFlow raises).
 * max_label_len <= 127, classes <= 256, frames up to ~15 000 (LDS); QK_ERR_UNSUPPORTED beyond.
 * Workspace: qk_ctc_workspace_bytes (the alpha and beta lattices, 2 x frames x (2 max_label_len + 1) floats per sample). */
size_t qk_ctc_workspace_bytes(int32_t batch, int32_t frames, int32_t max_label_len);
int qk_ctc_batch_cost(int32_t dtype, int32_t batch, int32_t frames, int32_t classes, const void *y_pred, const int32_t *labels,
                      int32_t max_label_len, const int32_t *input_length, const int32_t *label_length, float *cost, void *dy_pred,
                      void *workspace, size_t workspace_bytes, void *stream);

/* Softmax over the last axis of a (rows, cols <= 64) matrix, one wave per row -- the activation of the model's
 * TimeDistributed(Dense(62, activation='softmax')) output layer (models/interspeech_model.py:171-175) and its autodiff:
 *   fwd   y = softmax(logits + bias)          logits: fp32 (the GEMM's fp32 output), bias: fp32 or NULL, y: `dtype`
 *   bwd   dlogits = y * (dy - sum_j dy_j y_j)   (dtype; the operand of the layer's two gradient GEMMs)
 *         dbias[j] += sum over rows of dlogits  (fp32, ACCUMULATED: the caller zeroes it or hands in a gradient buffer; NULL = skip)
 * (Round 6: the model's output layer runs qk_dense_softmax_fwd / _bwd below; these two remain for callers that bring their own
 * logits -- the A/B composition behind QK_DBG_NO_FUSED_SOFTMAX.) */
int qk_softmax_rows_fwd(int32_t dtype, int64_t rows, int32_t cols, const float *logits, const float *bias, void *y, void *stream);
int qk_softmax_rows_bwd(int32_t dtype, int64_t rows, int32_t cols, const void *y, const void *dy, void *dlogits, float *dbias, void *stream);
/* The whole output layer as ONE launch per direction (round 6; csrc/qk_out_layer.hip) -- keras Dense(units, activation='softmax') on
 * the last axis of a (rows, in_dim) 16-bit matrix with fp32 master weights (models/interspeech_model.py:171-175 of the reference:
 * TimeDistributed(Dense(62, activation='softmax')), 51 200 rows x 256 at B = 256):
 *   fwd   y = softmax(x kernel + bias)                       x, y: `dtype` (QK_BF16 / QK_F16); kernel (in_dim, units), bias (units) fp32, bias may be NULL
 *   bwd   dl = s * y * (dy - sum_j dy_j y_j);  dx = dl kernel^T  (dtype);   s = (dy_scale_dev ? *dy_scale_dev : 1) * dy_scale -- a DEVICE
 *         scalar (fp32; NULL = 1) times a host factor: lets a loss node hand over an unscaled d loss / d y and its upstream gradient
 *         separately (the batch mean of qk_ctc_batch_cost: no elementwise pass over dy in between); applied in fp32;
 *         dkernel += x^T dl,  dbias += column sums of dl     (fp32, ACCUMULATED: the caller zeroes them or hands in gradient buffers; NULL = skip)
 * The products run on v_mfma_f32_16x16x32 with fp32 accumulation on the 16-bit roundings of kernel and dl; the softmax sees fp32 logits.
 * Supported: in_dim in {64, 128, 256}, 2 <= units <= 64 and even, 16-bit dtypes (qk_dense_softmax_supported; QK_ERR_UNSUPPORTED
 * otherwise -- the caller composes the layer from its parts).  The backward's workspace (caller-owned, qk_dense_softmax_bwd_workspace_bytes
 * for the CURRENT device; not needed when dkernel and dbias are both NULL) holds one slab of gradient partials per workgroup; a second
 * small launch sums the slabs in a fixed order: no float atomics, bit-repeatable gradients without QK_DBG_DETERMINISTIC. */
int qk_dense_softmax_supported(int32_t dtype, int64_t rows, int32_t in_dim, int32_t units);
int qk_dense_softmax_fwd(int32_t dtype, int64_t rows, int32_t in_dim, int32_t units, const void *x, const float *kernel, const float *bias,
                         void *y, void *stream);
size_t qk_dense_softmax_bwd_workspace_bytes(int32_t dtype, int64_t rows, int32_t in_dim, int32_t units);
int qk_dense_softmax_bwd(int32_t dtype, int64_t rows, int32_t in_dim, int32_t units, const void *x, const float *kernel, const void *y,
                         const void *dy, void *dx, float *dkernel, float *dbias, const float *dy_scale_dev, float dy_scale,
                         void *workspace, size_t workspace_bytes, void *stream);
/* *out += sum_i a[i] * w[i]  (a: `dtype`, w / out: fp32): a linear functional of the model output as one launch. */
int qk_weighted_sum(int32_t dtype, int64_t n, const void *a, const float *w, float *out, void *stream);

/* Batched form of the 16-bit kernel re-layout every forward / backward-data call otherwise does for itself: for job i,
 * write into workspaces[i] what a call of operation ops[i] (QK_OP_FWD, or QK_OP_BWD_DATA / QK_OP_BWD) with descriptor
 * descs[i] and kernel w[i] would write at the start of its workspace -- ONE launch for up to 32 jobs.  Meant to run once
 * behind each optimiser step; the calls of the next training step then pass desc.ws_has_kernel = 1 with those workspaces
 * and launch nothing but their GEMM kernel.  16-bit descriptors only (QK_ERR_INVALID_ARG otherwise); a job whose cq or fq
 * is not a multiple of 16 is SKIPPED, not refused (multiples of 16 are zero-padded to the kernels' 32-channel granule: the
 * workspace holds taps x pad32(cq) x 4 x pad32(fq) 16-bit values + 256 bytes): its calls run the fp32-MFMA kernels, which read the compact kernel in
 * place and never look at the workspace, so a model that mixes on-path and off-path layers hands over all of them. */
int qk_conv_prep_kernels(int32_t n, const qk_conv_desc_t *const *descs, const int32_t *ops, const float *const *w,
                         void *const *workspaces, void *stream);

/* qk_adam_step(_zero_grad) with the model's l2 kernel regularisers folded in.  Keras adds  l2 * sum(w^2)  of every
 * regularised kernel to the loss (models/interspeech_model.py:63,68,173: kernel_regularizer=l2(d.l2)); its gradient
 * 2 * l2 * w is applied here as  g = grad * grad_scale + decay[i] * param[i]  (decay: one coefficient per element,
 * 2 * l2 on regularised kernels and 0 on biases / slopes; NULL = no term).  The backward kernels then remain the
 * only writers of the gradient buffer, which is what lets them add into it directly (QK_BWD_ACCUMULATE) and lets a
 * data-parallel caller send a bucket the moment its last kernel has finished.  zero_grad != 0 clears grad. */
int qk_adam_step_l2(float *param, float *grad, float *m, float *v, const float *decay, size_t n, float lr,
                    float beta1, float beta2, float eps, int32_t step, float grad_scale, int32_t zero_grad,
                    void *stream);

/* qk_adam_step_l2 with the step number ON THE DEVICE: *step_dev = number of steps applied so far (start it at 0).  The kernel
 * applies step *step_dev + 1 (Keras' bias-corrected rate is formed in the kernel, in double as on the host) and a one-thread
 * launch behind it increments the counter.  No launch argument depends on the step, so a training step captured as ONE graph
 * (forward, loss, backward, this call, qk_conv_prep_kernels) replays correctly; hand the same counter to the post-ops
 * (qk_postop_t.drop_seed_dev) and the dropout masks change from replay to replay as well.  Replaces the Keras optimizer's
 * host-side `iterations` variable (the reference trains through Model.fit: /root/reference/models/interspeech_model.py:106,184).
 * decay may be NULL. */
int qk_adam_step_dev(float *param, float *grad, float *m, float *v, const float *decay, size_t n, float lr,
                     float beta1, float beta2, float eps, int32_t *step_dev, float grad_scale, int32_t zero_grad,
                     void *stream);

#ifdef __cplusplus
}
#endif
#endif /* QK_H_ */
