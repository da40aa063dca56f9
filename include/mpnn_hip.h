/*
 * mpnn_hip.h -- C ABI of libmpnn_hip.so, the MI355X (gfx950) kernel library for
 * the multipath-nn training hot path.
 *
 * The reference (MasonMcGill/multipath-nn) has no FFI: its boundary is the
 * Python class protocol of scripts/lib/layer_types.py + scripts/lib/net_types.py
 * whose arithmetic is delegated to TensorFlow ops.  Each entry point below
 * replaces the TensorFlow op call sites it cites (paths relative to the
 * reference root).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless stated; activations NHWC fp32,
 *     filters HWIO fp32 (TensorFlow defaults, net_types.py:50-51);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*),
 *     allocates nothing, never synchronises and is hipGraph-capturable;
 *   - return value: 0 = launched, >0 = hipError_t, <0 = MPNN_E_* (bad shape);
 *     nothing throws across the ABI;
 *   - "pre-activation" buffers hold the conv sums BEFORE BatchNorm; BatchNorm
 *     (+ReLU) is applied by the consumer while loading (mpnn_act).
 */
#ifndef MPNN_HIP_H
#define MPNN_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define MPNN_MAX_NODES 128     /* routing-tree nodes (mpnn_route)            */
#define MPNN_MAX_SINKS 4       /* children of one switch                      */

#define MPNN_E_SHAPE   (-1)   /* unsupported geometry / channel count      */
#define MPNN_E_ARG     (-2)   /* inconsistent arguments                     */

/* BatchNorm statistics are accumulated with fp64 atomics into MPNN_BN_SLOTS
 * replicated slots (a workgroup adds to slot blockIdx % nslot) so that
 * thousands of workgroups do not serialise on one address; readers add the
 * slots.  A statistics buffer is [MPNN_BN_SLOTS][2*C] doubles of which the
 * first `nslot` are used (few slots for layers with few workgroups: every
 * consumer workgroup re-adds them in its prologue).  MPNN_BN_SLOTS is the CAPACITY callers size
 * the buffers for; every record carries the number actually used (mpnn_act.nslot, out_nslot,
 * red_nslot: the engine uses 8 everywhere, lib/_plan.py:_nslot -- measured best). */
#define MPNN_BN_SLOTS 16

#define MPNN_ACT_IDENTITY 0   /* raw values (pyramid input, gradients)      */
#define MPNN_ACT_BN_BATCH 1   /* relu(bn(x)) with batch statistics ('tr')   */
#define MPNN_ACT_BN_MOVING 2  /* relu(bn(x)) with moving averages  ('ev')   */
#define MPNN_ACT_RELU 3       /* relu(x): the `Rect` after a plain `Conv` (layer_types.py:76-79) */

/* An activation as its consumer sees it: pre-activation values plus the
 * BatchNorm + ReLU to apply on load.  Replaces BatchNorm.link / Rect.link /
 * MultiscaleBatchNorm.link / MultiscaleRect.link (layer_types.py:219-249,
 * 76-79, 196-199) and ToPyramid.link's strided pick (layer_types.py:118-125:
 * `shift` = log2 of the subsampling factor). */
typedef struct {
    const float  *x;       /* [n, H<<shift, W<<shift, C]                      */
    const double *sum;     /* [SLOTS][2*C]: sum, sum of squares over (n,H,W)  */
    const float  *gamma, *beta, *m_avg, *v_avg;   /* [C] each                 */
    float eps;             /* BatchNorm eps (layer_types.py:220)              */
    int   cnt;             /* n*H*W: element count behind `sum`               */
    int   C;
    int   shift;
    int   mode;            /* MPNN_ACT_*                                       */
    int   nslot;           /* slots of `sum` actually in use (1..MPNN_BN_SLOTS) */
} mpnn_act;

/* The activation MATERIALISED: y = relu(bn(x)) (identity mode: a copy), [n_pix, C] -- exactly what
 * the consumers compute while loading (same coefficients, same expression).  The hot path never
 * needs it (nothing post-ReLU is stored); callers that want a block's output tensor do, and the
 * parity tests read the ReLU decisions of the device from it.  C % 4 == 0, C <= 256, shift == 0. */
int mpnn_bn_relu_fwd(const mpnn_act *a, float *y, long n_pix, void *stream);

/* ---- weight packing ------------------------------------------------------
 * HWIO [3][3][Cin][Cout] -> forward pack [9][ceil(Cin/16)][4][Cout][4]
 * (k-interleaved so one 16-byte load feeds four v_mfma_f32_16x16x4_f32) and
 * transposed+flipped pack for the input-gradient convolution.
 * One launch packs `n_desc` tensors described by the device table `desc`
 * (6 ints each: src_off, fwd_off, bwd_off, Cin, Cout, reserved; offsets in
 * floats relative to `params` / `packs`). */
int mpnn_pack_weights(const float *params, float *packs, const int *desc,
                      int n_desc, void *stream);

/* First launch of a training step: mpnn_pack_weights plus clearing `zero_bytes` bytes at `zero`
 * (both multiples of 16; the step's accumulators: BatchNorm statistic slots, loss sums, gradient
 * tensor) in the same kernel -- one launch instead of a pack and two memsets. */
int mpnn_step_begin(const float *params, float *packs, const int *desc, int n_desc,
                    void *zero, long zero_bytes, void *stream);
/* (n_desc == 0: clear only -- the packs are kept current by mpnn_talr_momentum_step.) */

/* ---- multiscale conv block, forward --------------------------------------
 * One scale of MultiscaleConvMax.link (layer_types.py:181-185):
 *   out = bias + conv3x3_same(act(a)) [+ conv3x3_same(v)]
 * and, fused, the per-channel sum / sum-of-squares that BatchNorm's
 * tf.nn.moments needs (layer_types.py:232) and, if `pool_out` is given, the
 * 2x2/2 max-pool of `out` (the `pool(self.x[i-1])` of layer_types.py:185 that
 * the next coarser scale consumes: pooled ONCE, by the producer).  `v` is that
 * pooled PRE-BN map of the next finer scale, [n, H, W, Cv] (NULL for the first
 * scale). */
typedef struct {
    mpnn_act a;
    const float *v;  int Cv;
    const float *wa_pack;  const float *wv_pack;   /* forward packs           */
    const float *bias;                              /* [Cout]                  */
    float  *out;                                    /* [n, H, W, Cout]         */
    float  *pool_out;                               /* [n, H/2, W/2, Cout] or NULL */
    double *out_sum;                                /* [SLOTS][2*Cout], accumulated */
    int out_nslot;                                  /* slots of out_sum to spread over */
    int n, H, W, Cout;
    /* Routed evaluation (mpnn_msconv_fwd_group only; 'ev' mode, BatchNorm moving averages): when
     * `idx` is set, the launch processes the *cnt samples idx[0..*cnt) -- sample slot s is image
     * idx[s] of EVERY buffer of the record (a, v, out, pool_out): inputs are gathered and results
     * scattered by the tile loader / epilogue themselves, nothing is copied.  `cnt` is read on the
     * DEVICE when the kernel starts (written by an earlier mpnn_exit_ev on the same stream: no host
     * sync); `n` stays the capacity the grid is sized for.  The samples that did not reach this
     * node keep whatever the buffers held.  Replaces the 0/1 masks p_ev of net_types.py:127-131. */
    const int *idx;
    const int *cnt;
} mpnn_conv_fwd_args;
int mpnn_msconv_fwd(const mpnn_conv_fwd_args *args, void *stream);
/* Up to four mutually independent forward convs (one wavefront level of the block x scale grid)
 * as ONE launch; `args` is a HOST array of `count` records (sizes the grid) and `dev_args` a
 * DEVICE copy of it (read by the kernel; uploaded once per plan).  The grid is 1-D, holds exactly
 * the workgroups that have work and is fitted to what is resident at once, shared between the
 * members by work.  A single member on an 8x8 / 4x4 map with >= 64 input channels (multiples of
 * 32 per operand) in TRAINING mode (a.mode == MPNN_ACT_BN_BATCH) takes the K-split body (512
 * threads, two chunks of a 32-channel unit in parallel); results are the same up to fp32 summation
 * order.  Evaluation-mode launches never take it.  Members on maps with W % 16 == 0 whose input is
 * one to three 16-channel chunks -- or a 1/3-channel image plus a 16-channel operand V (block 0's
 * second scale) -- take the wave-per-strip bodies of conv_strip.h when the record's sample capacity
 * `n` is >= 512 (MPNN_STRIP: minimum, 0 = off), in training mode (batch statistics) as in evaluation
 * mode: the one-chunk body gives the same bits as mpnn_msconv_fwd, the multi-chunk / image + V bodies
 * the same values up to fp32 summation order.  Groups of 8x8 / 4x4 members with Cout % 32 == 0 in evaluation mode and
 * `n` >= 1024 (MPNN_FWD_WIDE) use 32-channel output tiles: the same bits.  Which body runs depends only on the shapes and on `n` -- never on the
 * grouping, the index list or the device-side count -- so a conv gives bit-identical rows on all
 * samples or on a routed sub-batch of the same capacity. */
int mpnn_msconv_fwd_group(const mpnn_conv_fwd_args *args, const mpnn_conv_fwd_args *dev_args, int count,
                          void *stream);
/* CO-TRAINING (no reference counterpart in one call: scripts/train-nets:81-88,159-164 trains the nets of an experiment one
 * after another, each at batch 128 -- one net cannot fill 256 compute units).  The `_rep` / `_multi` entry points run the
 * SAME launch of `reps` nets of one architecture as ONE grid: `args` / `dev_args` hold reps * count records, net r's at
 * [r * count, (r + 1) * count) -- identical shapes and modes, every pointer net r's own (its batch, its parameters, its
 * BatchNorm statistics).  Per net the arithmetic is that of the single-net launch on resident slots / reps. */
int mpnn_msconv_fwd_group_rep(const mpnn_conv_fwd_args *args, const mpnn_conv_fwd_args *dev_args, int count,
                              int reps, int share, void *stream);
/* (share: every net's grid is sized for resident slots / share; <= 0 means reps.  reps = 1 with share = K launches ONE net
 * with exactly the grid it has inside a joint launch of K.) */

/* ---- BatchNorm(+ReLU) backward pieces ------------------------------------
 * Backward of `y = relu(gamma * (s - m) / sqrt(v + eps) + beta)` THROUGH the
 * batch statistics (autodiff of layer_types.py:231-236):
 *   dz   = dy * [y > 0]
 *   red  = [sum dz, sum dz * xhat]                 (-> dbeta, dgamma)
 *   g    = gamma*rstd * (dz - red0/cnt - xhat*red1/cnt)
 * mpnn_bn_bwd_reduce computes dz and accumulates red; mpnn_bn_bwd_apply
 * turns dz into g in place once red is complete. */
typedef struct {
    const float *s;        /* pre-BN values [n,H,W,C]                          */
    mpnn_act bn;           /* the BatchNorm of s (x field unused)              */
    const double *red;     /* [SLOTS][2*C] sum dz, sum dz*xhat (NULL: zero)    */
    int red_nslot;         /* slots of red in use                              */
} mpnn_bn_ctx;
int mpnn_bn_bwd_reduce(const float *dy, const mpnn_bn_ctx *ctx, float *dz,
                       double *red_out, long n_pix, void *stream);
int mpnn_bn_bwd_apply(float *dz_inout, const mpnn_bn_ctx *ctx, long n_pix,
                      void *stream);

/* ---- multiscale conv block, input gradients --------------------------------
 * mpnn_msconv_dgrad_horz: dy = conv3x3_same^T(g, w_horz) [+ dy_extra], i.e. the
 * gradient w.r.t. the block's (post-ReLU) input at this scale, fused with the
 * producer's mpnn_bn_bwd_reduce (prev != NULL) or stored raw (prev == NULL).
 * mpnn_msconv_dgrad_vert: dv = conv3x3_same^T(g, w_vert) at the coarse scale,
 * routed through the 2x2 max-pool to the FIRST maximum of each window of the
 * finer pre-BN map and added to that map's BatchNorm backward:
 *   g_fine = bn_bwd_apply(dz_fine) + maxpool_bwd(dv)        (in place on dz_fine)
 * Autodiff of layer_types.py:181-185. */
typedef struct {
    const float *g;  int Cg;            /* [n,H,W,Cg] gradient w.r.t. pre-BN sums */
    const mpnn_bn_ctx *g_ctx;            /* non-NULL: `g` holds dz and the BatchNorm backward
                                            (mpnn_bn_bwd_apply) is applied while loading it */
    const float *w_pack;                 /* backward pack                          */
    const float *dy_extra;               /* [n,H,W,Cout] or NULL                   */
    const mpnn_bn_ctx *prev;             /* producer BatchNorm context or NULL     */
    float  *out;                         /* dz (prev != NULL) or dy                */
    double *red_out;                     /* [SLOTS][2*Cout] accumulated (prev != NULL) */
    int n, H, W, Cout;
    int accumulate;                      /* 1: out += result (a map feeding SEVERAL child blocks, tree nets:
                                            the ReLU mask and the reductions are linear in dy, so each child's
                                            launch adds its own masked share) */
} mpnn_dgrad_horz_args;
int mpnn_msconv_dgrad_horz(const mpnn_dgrad_horz_args *args, void *stream);

typedef struct {
    const float *g;  int Cg;            /* coarse gradient [n,H,W,Cg]             */
    const mpnn_bn_ctx *g_ctx;            /* as in mpnn_dgrad_horz_args              */
    const float *w_pack;                 /* backward pack of w_vert                */
    const mpnn_bn_ctx *fine;             /* BatchNorm context of the finer scale   */
    int fine_has_dz;                     /* 0: finer BN output has no consumer     */
    float *dz_g_fine;                    /* [n,2H,2W,Cout]: dz in, g out           */
    int n, H, W, Cout;                   /* H, W = COARSE size; Cout = fine chans  */
} mpnn_dgrad_vert_args;
int mpnn_msconv_dgrad_vert(const mpnn_dgrad_vert_args *args, void *stream);
/* Both input gradients of one scale (same g, independent outputs) as ONE launch. */
int mpnn_msconv_dgrad_pair(const mpnn_dgrad_horz_args *horz, const mpnn_dgrad_vert_args *vert,
                           void *stream);

/* ---- multiscale conv block, weight gradients -------------------------------
 * dW_horz = act(a)^T (*) g, dW_vert = v^T (*) g (v = the pooled finer map, as in
 * mpnn_conv_fwd_args), db = sum g.
 * The pixel range is divided over `n_split` workgroup rows; split s WRITES its
 * partial sums (plain stores, every element exactly once) to dwa/dwv/db +
 * s*split_stride.  With n_split == 1 those may be the gradient tensors
 * themselves; otherwise they point into a scratch slab and mpnn_slab_reduce
 * adds the splits in a fixed order (bitwise-reproducible gradients; fp32
 * atomics from hundreds of workgroups onto a few-KB tensor serialise). */
typedef struct {
    mpnn_act a;
    const float *v;  int Cv;
    const float *g;                      /* [n,H,W,Cout]                           */
    const mpnn_bn_ctx *g_ctx;            /* as in mpnn_dgrad_horz_args              */
    float *dwa;  float *dwv;  float *db; /* HWIO partial sums of split 0, [Cout]   */
    long split_stride;                   /* floats between consecutive splits      */
    int n, H, W, Cout;
    int n_split;
} mpnn_wgrad_args;
int mpnn_msconv_wgrad(const mpnn_wgrad_args *args, void *stream);
/* Everything the backward pass does with g of one scale -- dgrad-horz (NULL: none), dgrad-vert
 * (NULL: none) and the weight gradients -- as ONE launch of independent workgroups. */
int mpnn_msconv_bwd_scale(const mpnn_dgrad_horz_args *horz, const mpnn_dgrad_vert_args *vert,
                          const mpnn_wgrad_args *wgrad, void *stream);
/* One DEPENDENCY LEVEL of the backward pass as one launch: g(b, S) (block b, absolute scale S) is final once
 * the child block's dgrad-horz B(b+1, S) and the coarser scale's dgrad-vert B(b, S+1) have run, so the
 * mpnn_msconv_bwd_scale triples with equal (blocks from the end) + (scales from the coarsest) are mutually
 * independent (autodiff of layer_types.py:181-185; the reference leaves the order to TensorFlow's executor,
 * net_types.py:35).  Members must not write the same output map (tree nets: siblings that accumulate into one
 * parent map go to different launches).  The CALLER budgets the workgroups: wg_horz / wg_vert = workgroups per
 * 16-channel tile row of the two input-gradient bodies, wgrad->n_split = pixel split of the weight gradients;
 * mpnn_msconv_bwd_level_slots gives the workgroups that are resident at once for a set of member shapes
 * (H, W, Cout of g), or MPNN_E_SHAPE when no kernel variant covers the set.
 * The member records live in DEVICE memory: _prepare fills `count` records of _record_size() bytes in host
 * memory, the caller copies them to the device once per plan and passes the device pointer to every launch. */
#define MPNN_BWD_LEVEL_MAX 4
typedef struct {
    const mpnn_dgrad_horz_args *horz;    /* NULL: none */
    const mpnn_dgrad_vert_args *vert;    /* NULL: none */
    const mpnn_wgrad_args *wgrad;
    int wg_horz, wg_vert;
} mpnn_bwd_member;
int mpnn_msconv_bwd_level_record_size(void);
int mpnn_msconv_bwd_level_slots(const int *H, const int *W, const int *Cout, int count);
int mpnn_msconv_bwd_level_prepare(const mpnn_bwd_member *members, int count, void *host_records);
int mpnn_msconv_bwd_level(const mpnn_bwd_member *members, int count, const void *dev_records, void *stream);
/* One dependency level of `reps` co-trained nets (see mpnn_msconv_fwd_group_rep): reps * count members / records, net r's
 * at [r * count, ...); the same shapes and the same workgroup budgets in every net (budget against _slots / reps). */
int mpnn_msconv_bwd_level_prepare_rep(const mpnn_bwd_member *members, int count, int reps, void *host_records);
int mpnn_msconv_bwd_level_rep(const mpnn_bwd_member *members, int count, int reps, const void *dev_records, void *stream);
/* Number of 64-pixel tiles (upper bound of n_split) for a map, or MPNN_E_SHAPE. */
int mpnn_wgrad_tiles(int n, int H, int W);
/* dst[i] = sum_{s<n_split} src[s*stride + i].  table: 6 ints per work item:
 * src_off (floats in `slabs`), dst_off (floats in `grads`), count (<= MPNN_SLAB_ITEM),
 * n_split, stride, reserved.  One workgroup per item; a larger count is MPNN_E_ARG-free
 * undefined behaviour, so split larger tensors into several items. */
#define MPNN_SLAB_ITEM 1024
int mpnn_slab_reduce(const float *slabs, float *grads, const int *table, int n_items,
                     void *stream);

/* ---- single-scale Conv (scripts/lib/layer_types.py:55-74) ---------------------
 * y = b + conv2d_same(act(x), w) with supp x supp filters, supp = 3 (the direct 3x3 MFMA body of
 * the multiscale path) or supp = 1 (a GEMM over pixels, [n*H*W, Cin] x [Cin, Cout]).
 *   fwd  : w = the forward pack (supp 3, mpnn_pack_weights) or the HWIO tensor itself = [Cin][Cout] (supp 1)
 *   dgrad: dx = conv^T(g, w) [* (relu_src > 0): the producer's Rect]; w = the backward pack (supp 3) or
 *          [Cin][Cout] (supp 1); `scratch` (supp 3 with relu_src): 2*Cin doubles, overwritten
 *   wgrad: dw = act(x)^T (*) g, db = sum g.  supp 3: as mpnn_msconv_wgrad (n_split > 1: dw / db point into
 *          a slab, split_stride apart, then mpnn_slab_reduce); supp 1: ADDED into (caller-zeroed) dw / db.
 * Limits: supp 3 as the multiscale entry points; supp 1: Cin, Cout <= 256. */
typedef struct {
    mpnn_act a;  const float *w;  const float *bias;  float *out;
    int n, H, W, Cout;  int supp;
} mpnn_conv_nhwc_fwd_args;
int mpnn_conv_nhwc_fwd(const mpnn_conv_nhwc_fwd_args *args, void *stream);
typedef struct {
    const float *g;  int Cg;  const float *w;
    const float *relu_src;               /* [n,H,W,Cin] pre-activation of the producer, or NULL     */
    double *scratch;
    float *dx;
    int n, H, W, Cin;  int supp;
} mpnn_conv_nhwc_dgrad_args;
int mpnn_conv_nhwc_dgrad(const mpnn_conv_nhwc_dgrad_args *args, void *stream);
typedef struct {
    mpnn_act a;  const float *g;  float *dw;  float *db;
    long split_stride;  int n_split;
    int n, H, W, Cout;  int supp;
} mpnn_conv_nhwc_wgrad_args;
int mpnn_conv_nhwc_wgrad(const mpnn_conv_nhwc_wgrad_args *args, void *stream);

/* ---- exit head + router, first affine map ----------------------------------
 * y_k = flatten(act(a)) @ w_k + b_k [+ alpha_cpt * k_cpt[n] * w_k[K]] for up to
 * two weight sets sharing the input: the LogReg head's LinTrans and the
 * router's first LinTrans (layer_types.py:39-53 after Select(-1),
 * arch_and_hypers.py:45-49,66-70; the k_cpt column is net_types.py:149-160).
 *
 * The exit-path entry points are TABLE DRIVEN: `dev_table` is a DEVICE array
 * of `count` argument records (uploaded once per plan); one launch serves every
 * exit of the routing tree (the conv trunk does not depend on them). */
typedef struct {
    mpnn_act a;  int HW;                 /* input [n, HW, C] flattened to K=HW*C   */
    const float *w[2];  const float *b[2];  float *y[2];  int M[2];   /* M <= 16   */
    const float *k_cpt;  float alpha_cpt;  int extra_col[2];
    int n;
    /* Scratch of mpnn_lin_fwd_ks (both NULL: the record is never sliced; ignored by mpnn_lin_fwd):
     *   kpart: ceil(n/16) * MPNN_LIN_KSLICES * 512 floats, any contents
     *   kcnt : ceil(n/16) ints, ZERO before the first launch (left zero by every launch) */
    float *kpart;  int *kcnt;
} mpnn_lin_fwd_args;
#define MPNN_LIN_KSLICES 8
int mpnn_lin_fwd(const mpnn_lin_fwd_args *dev_table, int count, int n_max, void *stream);
/* The same map for SMALL batches (the training step: 8 row groups), K-sliced: a record with scratch and
 * K >= 512 is split over S = min(MPNN_LIN_KSLICES, K / 256) 256-thread workgroups per 16 rows; each leaves
 * its partial tile in kpart, and the LAST to arrive (ticket counter kcnt, which it resets) adds the S
 * partials in slice order -- the result does not depend on which workgroup that is (bit-identical from
 * launch to launch); it differs from mpnn_lin_fwd's in the last bits (another summation tree).
 * k_max >= HW*C of every record (sizes the grid). */
int mpnn_lin_fwd_ks(const mpnn_lin_fwd_args *dev_table, int count, int n_max, int k_max, void *stream);

typedef struct {
    mpnn_act a;  int HW;
    const float *w[2];  const float *dy[2];  int M[2];
    float *dw[2];  float *db[2];         /* written (sums over all n rows, fixed order) */
    float *dx;                           /* [n, HW*C] written, or NULL             */
    const float *k_cpt;  float alpha_cpt;  int extra_col[2];
    int n;
    /* Optional fusion of mpnn_bn_bwd_reduce for an exit whose block has no child (its dX is the
     * only gradient of that map): when dz_out is set (a.mode must be a BatchNorm mode), the kernel
     * also writes dz = dX where relu(bn(x)) > 0 else 0 to dz_out [n, HW*C] and adds
     * [sum dz, sum dz * xhat] per channel to red_out (fp64 [red_nslot][2C], caller-zeroed). */
    float *dz_out;  double *red_out;  int red_nslot;
    /* Scratch of mpnn_lin_bwd_rs (ignored by mpnn_lin_bwd):
     *   kpart: ceil((K+1)/64) * MPNN_LIN_RSPLIT * MPNN_LIN_RS_TILE floats, any contents
     *   kcnt : ceil((K+1)/64) ints, ZERO before the first launch (left zero by every launch) */
    float *kpart;  int *kcnt;
} mpnn_lin_bwd_args;
#define MPNN_LIN_RSPLIT 8
#define MPNN_LIN_RS_TILE 2080            /* 64 features x 32 outputs of dW + 32 of db */
int mpnn_lin_bwd(const mpnn_lin_bwd_args *dev_table, int count, int n_max, int k_max,
                 void *stream);
/* The same for SMALL batches, ROW-SPLIT: min(MPNN_LIN_RSPLIT, ceil(n_max / 64)) workgroups share the rows
 * of a 64-feature block; their dW / db partial tiles meet in kpart, added in row-group order by the last
 * to arrive (ticket counter kcnt) -- bit-identical from launch to launch.  Every record needs scratch. */
int mpnn_lin_bwd_rs(const mpnn_lin_bwd_args *dev_table, int count, int n_max, int k_max,
                    void *stream);

/* ---- exit tail: everything after the first affine map ----------------------
 * Head: Softmax + CrossEntropyError (layer_types.py:81-84, 262-272).
 * Router: BN -> ReLU -> LinTrans(R) -> BN -> ReLU -> LinTrans(n_sinks)
 * (arch_and_hypers.py:47-49; BatchNorm over the batch, layer_types.py:219-239).
 * One workgroup per exit owns the whole batch (batch statistics need every
 * sample; the batch is held in LDS).  Limits: n <= 128 per launch, n_cls <= 16,
 * R <= 16, n_sinks <= 4 (MPNN_E_SHAPE beyond). */
typedef struct {
    /* head (z == NULL: no head at this exit) */
    const float *z;  const float *y;  int n_cls;  float eps_ce;
    float *c_err;  float *d_cor;                      /* [n] each               */
    /* router (h1 == NULL: node has no router) */
    const float *h1;  int R;  int n_sinks;            /* [n,R] pre-BN           */
    const float *g1, *b1;  float *m1, *v1;            /* BN1 gamma/beta/avgs    */
    const float *w2, *bias2;                          /* [R,R], [R]             */
    const float *g2, *b2;  float *m2, *v2;
    const float *w3, *bias3;                          /* [R,n_sinks], [n_sinks] */
    float *h2;  float *r;  int r_stride;              /* [n,R] pre-BN2, [n,r_stride] */
    float *bn_save;                                   /* [4*R] batch m1,rstd1,m2,rstd2 */
    float bn_eps, bn_decay;
    int mode;                                         /* MPNN_ACT_BN_BATCH / _MOVING */
    int n;
    /* Optional: accumulators of the launch that FOLLOWS (mpnn_route adds the TALR node statistics and the loss
     * sums with atomics): this record's head workgroup clears clear_f[0 .. n_clear_f) and clear_d[0 .. n_clear_d).
     * A training step then needs no clearing launch (the BatchNorm slot sums are cleared by their last reader,
     * mpnn_backward_finish / mpnn_bn_finalize). */
    float *clear_f;  int n_clear_f;  double *clear_d;  int n_clear_d;
    /* Width of the SECOND hidden layer (0: the same as R).  w2 is [R, R2], bias2 / g2 / b2 / m2 / v2 [R2], w3
     * [R2, n_sinks], h2 [n, R2], bn_save [2 R + 2 R2] = m1, rstd1, m2, rstd2.  The tuned kernels need R2 == R <= 16
     * and n_cls <= 16; the any-width forms below (mpnn_*_gen) take anything up to mpnn_exit_gen_check's limits. */
    int R2;
    /* Optional (K training steps in ONE hipGraph, lib/_plan.py:run_steps): this record's head workgroup copies the
     * MPNN_HYP_N schedule values hyp_src[0 .. MPNN_HYP_N) to hyp_dst -- the buffer mpnn_route and the optimizer of THIS
     * step read -- so that step j of the graph runs with the values staged in slot j, without a launch of its own. */
    const float *hyp_src;  float *hyp_dst;
    /* The SECOND router BatchNorm's own hyper-parameters (bn_eps / bn_decay above are the first one's: the two are
     * separate BatchNorm(d, eps) layers in the router chain, arch_and_hypers.py:47-49).  Always read: set both. */
    float bn_eps2, bn_decay2;
} mpnn_exit_tail_args;
int mpnn_exit_tail_fwd(const mpnn_exit_tail_args *dev_table, int count, int n_max,
                       void *stream);

typedef struct {
    mpnn_exit_tail_args f;
    const float *w_cerr;                 /* [n] dL/dc_err                          */
    const float *dr;                     /* [n,r_stride] dL/dr                     */
    float *dz;                           /* [n,n_cls] dL/dz (head logits)          */
    float *dh1;                          /* [n,R] dL/dh1                           */
    float *dg1, *db1, *dw2, *dbias2, *dg2, *db2, *dw3, *dbias3;   /* written      */
    float *dh2;                          /* [n,R2] scratch (mpnn_exit_tail_bwd_gen: dL/dh2; the tuned kernel ignores it) */
} mpnn_exit_tail_bwd_args;
int mpnn_exit_tail_bwd(const mpnn_exit_tail_bwd_args *dev_table, int count, int n_max,
                       void *stream);

/* ---- one exit in evaluation mode, with routing ------------------------------
 * Head + router of one tree node in 'ev' mode (moving-average BatchNorm:
 * layer_types.py:237-238), any batch size, and the node's hard routing decision
 * pi_ev = one-hot(arg-max r) (net_types.py:127-129, first index on ties) turned
 * into per-child sample lists:
 *   z = flatten(act(a)) @ w_head + b_head -> softmax -> c_err, d_cor   (as mpnn_exit_tail_fwd)
 *   r = router MLP (as mpnn_lin_fwd + mpnn_exit_tail_fwd with MPNN_ACT_BN_MOVING)
 *   sample s (image idx[s]) is APPENDED to child_idx[i] (count child_cnt[i], atomically) for
 *   i = arg-max r, when sink i has a list (a child block); sinks without one (the exit's own
 *   classifier leaf) need none.
 * The sample list idx[0..*cnt) is the node's own (NULL: all n samples); c_err, d_cor and r are
 * indexed by IMAGE (r: [image][r_stride]), entries of samples that do not reach the node are
 * not touched (the caller clears them once per batch).  Table driven like the other exit-path
 * entry points; the grid is sized for n_max samples, counts are read on the device.
 * Limits (mpnn_exit_ev_check, host records): n_cls <= 16, R <= 16, n_sinks <= 4, C <= 128,
 * HW*C % 16 == 0. */
typedef struct {
    mpnn_act a;  int HW;                 /* the block's coarsest scale, pre-BN: [images, HW, C] */
    const float *w_head, *b_head;  int n_cls;         /* NULL: no classifier at this node */
    const float *y;  float eps_ce;  float *c_err, *d_cor;
    const float *w1, *b1;  int R;  int n_sinks;       /* NULL: no router */
    int extra_col;  const float *k_cpt;  float alpha_cpt;   /* dyn_k_cpt column (net_types.py:149-160) */
    const float *g1, *be1, *m1, *v1;  const float *w2, *bias2;
    const float *g2, *be2, *m2, *v2;  const float *w3, *bias3;
    float bn_eps;
    float *r;  int r_stride;
    const int *idx;  const int *cnt;  int n;           /* this node's sample list / capacity */
    int *child_idx[MPNN_MAX_SINKS];  int *child_cnt[MPNN_MAX_SINKS];
    int R2;                              /* width of the second hidden layer (0: R); see mpnn_exit_tail_args */
    float *z, *h1;                       /* scratch of mpnn_exit_ev_gen: [n, n_cls] head logits, [n, R] first router map
                                          * (rows by IMAGE; the tuned mpnn_exit_ev keeps both in LDS and ignores them) */
    float bn_eps2;                       /* epsilon of the SECOND router BatchNorm (bn_eps: the first one's); always read */
} mpnn_exit_ev_args;
int mpnn_exit_ev(const mpnn_exit_ev_args *dev_table, int count, int n_max, void *stream);
int mpnn_exit_ev_check(const mpnn_exit_ev_args *host_record);

/* The routed evaluation's dense prefix made routed after the fact (csrc/exit_ev.hip): the exits of the blocks whose
 * convs run on every sample anyway (the first d0 tree depths) are evaluated densely in ONE mpnn_exit_ev launch; this
 * launch then walks every sample through the prefix's switches (arg-max of the router outputs, first index on ties:
 * net_types.py:127-129), clears r / c_err / d_cor of the prefix nodes the sample does not reach and appends it to the
 * sample list of the frontier block it arrives at.  Same results as the exit-by-exit routed pass, (d0 - 1) serial
 * launches fewer.  Records in topological order; parent[j] = record of the nearest switch above record j (-1: every
 * sample reaches it), parent_sink[j] = the sink of that switch that leads to j; n_sinks[j] = 0: no router at j;
 * c_err[j] / d_cor[j] NULL: no head at j.  front_*: the lists to fill.  host_rec is validated, dev_rec (the same
 * record in device memory) is what the kernel reads. */
#define MPNN_PREFIX_MAX 64
typedef struct {
    int n, count, n_front, pad_;
    int parent[MPNN_PREFIX_MAX], parent_sink[MPNN_PREFIX_MAX], n_sinks[MPNN_PREFIX_MAX], r_stride[MPNN_PREFIX_MAX];
    float *r[MPNN_PREFIX_MAX];
    float *c_err[MPNN_PREFIX_MAX];
    float *d_cor[MPNN_PREFIX_MAX];
    int front_parent[MPNN_PREFIX_MAX], front_sink[MPNN_PREFIX_MAX];
    int *front_idx[MPNN_PREFIX_MAX];
    int *front_cnt[MPNN_PREFIX_MAX];
} mpnn_ev_prefix_args;
int mpnn_ev_prefix_walk(const mpnn_ev_prefix_args *host_rec, const mpnn_ev_prefix_args *dev_rec, void *stream);

/* ---- any-WIDTH forms of the exit path (csrc/exit_gen.hip) -------------------
 * LinTrans takes any n_chan (layer_types.py:39-53) and the router MLP any hidden width (arch_and_hypers.py:14,45-49);
 * the tuned kernels above hold n_cls <= 16 and two EQUAL hidden layers of <= 16 units.  The same argument records go
 * to these plain kernels (a thread per output element, intermediates recomputed from h1 / h2 / bn_save, no scratch, any
 * batch size; not latency-tuned) for n_cls <= 1024, R, R2 <= 256, C <= 256, HW * C <= 4096 (mpnn_exit_gen_check):
 *   mpnn_lin_fwd_gen        == mpnn_lin_fwd
 *   mpnn_lin_bwd_gen        == mpnn_lin_bwd with dx (or NULL); the dz_out fusion is not offered: the caller runs
 *                              mpnn_bn_bwd_reduce on dx instead
 *   mpnn_exit_tail_fwd_gen  == mpnn_exit_tail_fwd (batch statistics or, with mode MPNN_ACT_BN_MOVING, the averages)
 *   mpnn_exit_tail_bwd_gen  == mpnn_exit_tail_bwd
 *   mpnn_exit_ev_gen        == mpnn_exit_ev (children's lists appended with one atomic per sample) */
int mpnn_lin_fwd_gen(const mpnn_lin_fwd_args *dev_table, int count, int n_max, void *stream);
int mpnn_lin_bwd_gen(const mpnn_lin_bwd_args *dev_table, int count, int n_max, int k_max, void *stream);
int mpnn_exit_tail_fwd_gen(const mpnn_exit_tail_args *dev_table, int count, int n_max, void *stream);
int mpnn_exit_tail_bwd_gen(const mpnn_exit_tail_bwd_args *dev_table, int count, int n_max, void *stream);
int mpnn_exit_ev_gen(const mpnn_exit_ev_args *dev_table, int count, int n_max, void *stream);
int mpnn_exit_gen_check(int C, int K, int n_cls, int R, int R2, int n_sinks);

/* ---- the router: routing probabilities, costs and their gradients ----------
 * Replaces ActorNet._route/_route_sinks_dyn + cost assembly
 * (net_types.py:108-131,165-177), CriticNet's (net_types.py:193-243,273-280)
 * and SRNet's loss (net_types.py:93-95).  One thread per sample walks the
 * routing tree given as a device table `nodes` (8 ints per node, DFS preorder:
 * parent, sink_index, n_sinks, switch_id (-1: none), leaf_id (-1: none),
 * n_leaves, depth (root = 0), rank of the node when all nodes are sorted by
 * (depth, preorder index) -- the kernel walks the tree one LEVEL at a time), `sw_children` [n_switches][max_sinks] node
 * ids, and `node_ops` (n_ops + router.n_ops per node).
 *   outputs: p_tr, p_ev [n_nodes][n]; w_cerr [n_leaves][n] = dL/dc_err;
 *            dr [n_switches][n][max_sinks] = dL/dr; node_stat [n_nodes][2] =
 *            sum p_tr, sum p_tr^2 (TALR, net_types.py:25-27), accumulated;
 *            loss[4] accumulated sums: c_err, c_cpt, c_dec|c_cre, sample count.
 * hyp = device floats indexed by MPNN_HYP_*. */
#define MPNN_NET_SR     0
#define MPNN_NET_ACTOR  1
#define MPNN_NET_CRITIC 2
#define MPNN_HYP_LR   0
#define MPNN_HYP_MU   1
#define MPNN_HYP_TAU  2
#define MPNN_HYP_EPS  3
#define MPNN_HYP_KCPT 4
#define MPNN_HYP_KDEC 5
#define MPNN_HYP_KCRE 6
#define MPNN_HYP_ARTR 7
#define MPNN_HYP_N    16
typedef struct {
    int net_type;  int n_nodes, n_leaves, n_switches, max_sinks;
    int optimistic, use_cls_err, want_grad;
    const int *nodes;  const int *sw_children;  const float *node_ops;
    const float *hyp;
    const float *k_cpt_vec;              /* [n] per-sample k_cpt or NULL           */
    const float *r;                      /* [n_switches][n][max_sinks]             */
    const float *c_err;  const float *d_cor;          /* [n_leaves][n]             */
    float *p_tr, *p_ev;                  /* [n_nodes][n]                           */
    float *w_cerr;                       /* [n_leaves][n]                          */
    float *dr;                           /* [n_switches][n][max_sinks]             */
    float *node_stat;                    /* [n_nodes][2] accumulated               */
    double *loss;                        /* [4] accumulated                        */
    int n;  int n_total;                 /* n_total: samples the mean is over      */
    /* node_stat with MORE THAN TWO workgroups (two fp32 atomics onto a cleared sum commute; three do not): every
     * workgroup stores its partial sums in stat_part (floats [workgroups][n_nodes][2], then doubles [workgroups][4] for
     * the loss sums; room for ceil(n / 16) workgroups: ceil(n / 16) * (2 * n_nodes + 8) floats, 8-byte aligned), the last
     * one to finish (stat_ticket: one int, zero between launches) adds them up in workgroup order -- the TALR statistics,
     * and with them the whole step, are the same bits from run to run.  NULL: atomics (evaluation; small batches). */
    float *stat_part;  int *stat_ticket;
} mpnn_route_args;
int mpnn_route(const mpnn_route_args *args, void *stream);
/* mpnn_route of `count` co-trained nets with the same tree shape and batch size as one launch (host_table sizes it,
 * dev_table = the same records in device memory). */
int mpnn_route_multi(const mpnn_route_args *host_table, const mpnn_route_args *dev_table, int count, void *stream);

/* On-device compaction of the 'ev' sub-batch that reaches a node: indices of
 * samples with p_ev > 0, in order (wave64 ballot + prefix sum), and their
 * count -- consumed by later launches without a host sync. */
int mpnn_compact_by_branch(const float *p_ev, int n, int *idx_out, int *count_out,
                           void *stream);

/* ---- BatchNorm epilogue of a training step ----------------------------------
 * Moving averages (layer_types.py:233-234) from the forward sums and
 * dgamma / dbeta from the backward reductions, for every conv BatchNorm in one
 * launch.  table: 8 ints per BN: sum_off (doubles; same offset in `reds`; each
 * a [SLOTS][2*C] block of which table[7] slots are in use),
 * mavg_off, vavg_off (floats in `state`), C, pixels per image, gamma_goff,
 * beta_goff (floats in `grads`), nslot.  gamma_goff = -1 marks a BatchNorm whose output
 * nobody consumes: no gradients and -- as in the reference, whose tf.assign of the
 * averages only executes when the output is needed -- no moving-average update.
 * sums_keep (NULL: off): this launch is the LAST reader of the step's slot sums; with sums_keep set it copies
 * the forward sums there (same layout: the step's batch statistics stay inspectable) and leaves `sums` and
 * `reds` CLEARED, so that the next training step needs no clearing launch. */
int mpnn_bn_finalize(double *sums, double *reds, float *state, float *grads,
                     const int *table, int n_bn, float decay, int n_img, double *sums_keep, void *stream);

/* ---- TALR + L2 + momentum (net_types.py:24-37, tf.train.MomentumOptimizer) --
 * For every trainable element: g = grad + 2*k_l2*pbar_node*(w - w_eq)  (w_eq: the identity part of a
 * `res` layer, layer_types.py:46,52,65-72: seg[5] = offset of the tensor's w_eq in `w_eq`, or -1 = zero);
 * g *= s_node (* alpha_rtr for router params), s_node = 1/sqrt(mean p_tr^2);
 * accum = mu*accum + g; w -= lr*accum.  seg table: MPNN_SEG_INTS ints per work item
 * (offset, count, node, is_router, l2_bits (float as int), w_eq offset or -1,
 *  then for 3x3 conv weights: offset of the tensor in `params`, Cin, Cout, forward-pack offset,
 *  backward-pack offset (-1: none) in `packs`; Cin = 0 for every other tensor; reserved).
 * packs != NULL: every updated conv weight is also written to its slots of the weight packs
 * (mpnn_pack_weights' layout), so the next step needs no packing launch.
 * grad_scale multiplies the raw gradients (1/world_size after an all-reduce
 * sum); inv_n = 1 / (samples behind node_stat). */
int mpnn_talr_momentum_step(float *params, float *accum, const float *grads,
                            const int *seg, int n_seg, const float *node_stat,
                            const float *hyp, int talr, float inv_n, float grad_scale,
                            const float *w_eq, float *packs, void *stream);
#define MPNN_SEG_INTS 12

/* ---- training-batch assembly (scripts/lib/data.py:10-34) -------------------
 * x_out[i] = rand_shift(rand_flip(x_src[j_i])), y_out[i] = y_src[j_i] for a dataset resident in
 * device memory ([N,H,W,C] and [N,n_cls] fp32, C <= 4).  `draw` is a device array [n][4] of ints
 * (j, flip, du, dv) drawn on the host with the reference's call sequence on numpy's global
 * stream (randint(0, N); rand() < 0.5 for symmetric classes; randint(-r, r + 1, 2)).
 * out[u][v] = a[u + du][v + dv] where that exists, else the image's per-channel mean (computed
 * in fp64); flip mirrors v.  y_src / y_out may be NULL. */
int mpnn_augment_batch(const float *x_src, const float *y_src, const int *draw,
                       float *x_out, float *y_out, int n, int H, int W, int C, int n_cls,
                       void *stream);
/* The batches of `count` consumers (co-trained nets) from one dataset in ONE launch: consumer r's record buffer and
 * destinations in dev_table[r] (device memory), n images each. */
typedef struct { const int *draw;  float *x_out;  float *y_out; } mpnn_augment_dst;
int mpnn_augment_batch_multi(const float *x_src, const float *y_src, const mpnn_augment_dst *dev_table, int count,
                             int n, int H, int W, int C, int n_cls, void *stream);

/* mpnn_slab_reduce and mpnn_bn_finalize in ONE launch (same arguments and semantics; the two are
 * independent of each other and both end the backward pass). */
int mpnn_backward_finish(const float *slabs, float *grads, const int *slab_table, int n_items,
                         double *sums, double *reds, float *state, const int *bn_table,
                         int n_bn, float decay, int n_img, double *sums_keep, void *stream);

/* mpnn_backward_finish AND mpnn_talr_momentum_step in ONE launch (single-process training: nothing sits between the
 * gradients and their use; under data parallelism the all-reduce does, and the two stay separate).  Every slab item's
 * workgroup applies the update to the elements it has just reduced (item_seg: one MPNN_SEG_INTS row per slab item, same
 * fields as the optimizer's work items, count <= MPNN_SLAB_ITEM), every BatchNorm's workgroup to its gamma / beta
 * (bn_opt: 4 ints per record of bn_table: tree node, l2 bits of gamma, l2 bits of beta, reserved), and n_plain further
 * workgroups run the optimizer work items plain_seg whose gradients are already final in `grads`.  Element for element
 * the arithmetic of the two launches it replaces. */
int mpnn_backward_finish_opt(const float *slabs, const int *slab_table, int n_items, const int *item_seg,
                             double *sums, double *reds, float *state, const int *bn_table, int n_bn,
                             const int *bn_opt, float decay, int n_img, double *sums_keep,
                             float *params, float *accum, float *grads, const float *node_stat,
                             const float *hyp, int talr, float inv_n, float grad_scale, const float *w_eq,
                             float *packs, const int *plain_seg, int n_plain, void *stream);

/* mpnn_backward_finish_opt of `count` co-trained nets as one launch: one record per net (the arguments of the single-net
 * call), in host memory to size the launch and in device memory for the kernel. */
typedef struct {
    const float *slabs;  const int *slab_table;  int n_items;  const int *item_seg;
    double *sums, *reds;  float *state;  const int *bn_table;  int n_bn;  const int *bn_opt;  int n_img;  double *sums_keep;
    float *params, *accum, *grads;  const float *node_stat, *hyp;  int talr;  float inv_n, grad_scale;
    const float *w_eq;  float *packs;  const int *plain_seg;  int n_plain;
} mpnn_finish_net;
int mpnn_backward_finish_opt_multi(const mpnn_finish_net *host_table, const mpnn_finish_net *dev_table, int count,
                                   float decay, void *stream);

/* Workgroups of the mpnn_msconv_bwd_scale kernel (the variant for this shape, with or without a
 * dgrad-vert body) for an H x W x Cout scale that are resident on the
 * device at once (occupancy x compute units; needs a GPU).  The caller gives the weight-gradient
 * split (n_split x channel chunks x cout groups workgroups) about half of them, so that the dgrad
 * and wgrad workgroups of the launch all start together.  dgrad_items = 64-pixel tiles x 16-channel rows of the
 * two input-gradient bodies (the most workgroups they can use; 64-channel layers whose input gradients cannot fill
 * one workgroup per CU are limited to two workgroups per CU).  Negative = MPNN_E_*. */
int mpnn_msconv_bwd_scale_slots(int H, int W, int Cout, int has_dgrad, int has_vert, int dgrad_items);

/* Profiling aid (no reference counterpart).  Installs (or, with NULL, removes) a device buffer of
 * MPNN_TRACE_SLOTS (12) uint64 per workgroup of the largest grid to be traced: thread 0 of every
 * workgroup of the conv / dgrad / wgrad kernels stamps the 100 MHz device clock at its phase
 * boundaries (entry, tables ready, first tile staged, first unit done, loop done, exit; slot 6 = body
 * kind, slot 7 = units; slots 8-10: first unit's MFMAs done, next unit staged, epilogue done).  tools/trace_phases.py prints the timeline.  Synchronises the device. */
int mpnn_debug_set_trace(unsigned long long *buf);

/* ---- MaxPool / GlobalMaxPool of the single-scale layer family (scripts/lib/layer_types.py:86-100; no shipped spec
 * uses them) on NHWC fp32 maps.  MaxPool: tf.nn.max_pool(..., 'SAME') with window `win` and step `step` -- the
 * reference passes its hypers as (strides, k_shape), i.e. window = hypers.stride, step = hypers.supp (:90-94);
 * out = ceil(H / step) x ceil(W / step); backward: each window's gradient goes to its FIRST maximum in row-major order,
 * overlapping windows add (a gather per input element: no atomics).  global != 0: tf.reduce_max over H x W ->
 * y [n][C], cnt [n][C] = number of maxima (forward output), backward dx = [x == y] * dy / cnt.  The caller pools
 * PRE-activation maps (max-pool commutes with the ReLU its consumers apply on load). */
int mpnn_maxpool_fwd(const float *x, float *y, float *cnt, int n, int H, int W, int C, int win, int step,
                     int global, void *stream);
int mpnn_maxpool_bwd(const float *x, const float *y, const float *cnt, const float *dy, float *dx, int n, int H,
                     int W, int C, int win, int step, int global, void *stream);

/* Host function (no device work, no stream): the augmentation draws of scripts/lib/data.py:24-34 -- per sample
 * randint(0, n_src); rand() < 0.5 if sym[j] (sym == NULL: every class is symmetric); randint(-r_shift, r_shift + 1, 2)
 * -- replayed over consecutive 32-bit outputs of numpy's legacy MT19937 stream with numpy's own bounded-integer and
 * random_sample algorithms.  draw: [n][4] = (j, flip, du, dv), the record mpnn_augment_batch reads.  RESUMABLE so that
 * the stream is never over-drawn: `state` (4 longs; [0..2] zero to start a batch, [3] = 1 if the caller vouches that
 * every sym[] entry is set, which tightens the lower bound) carries the position inside the batch; a
 * call consumes ALL n_raw words it is given and returns the minimum number of words the remaining draws need (0: the
 * batch is complete); the caller hands over exactly that many next time (first call: n_raw = 0).  MPNN_E_ARG for bad
 * arguments or more words than the minimum. */
long mpnn_draw_augmentation(const unsigned int *raw, long n_raw, int n, long n_src, const unsigned char *sym,
                            int r_shift, int *draw, long *state);

/* The same draws from a PRIVATE MT19937 state in numpy's legacy layout (key[624], *pos in 0..624: the fields of
 * numpy.random.RandomState.get_state()), the generator stepped here: `batches` batches of n samples per call, written
 * to draw as [batches][n][4] -- or, with draw == NULL, only skipped.  A net trained beside others (lib/_co.py, train-nets
 * --shard-nets) gets the batches of the reference's serial experiment loop (scripts/train-nets:159-164: ONE global
 * stream, net after net) by advancing the experiment's stream over the iterations of the nets in front of it.  key / pos
 * are updated in place.  0, or MPNN_E_ARG. */
long mpnn_draw_augmentation_mt(unsigned int *key, int *pos, long batches, int n, long n_src, const unsigned char *sym,
                               int r_shift, int *draw);

/* Data parallelism (new here: the reference is single-process, scripts/train-nets:159-164).  Compute units every
 * persistent grid launched AFTER this call leaves free (0: none, the default): RCCL's all-reduce kernels run beside
 * the backward launches of the bucket sections (lib/_plan.py), and a grid fitted to every resident workgroup slot
 * then holds workgroups that only start when others exit.  Affects the grids the launchers choose and the slot counts
 * mpnn_msconv_bwd_scale_slots / mpnn_msconv_bwd_level_slots report.  Host-side state, not stream-ordered.  Returns the
 * previous value; a negative argument only queries. */
int mpnn_set_reserved_cus(int cus);

/* Measurement aids (no reference counterpart).  mpnn_debug_spin: `wgs` workgroups of `threads` threads that do nothing
 * but hold their slots for `us` microseconds (100 MHz device clock) -- a stand-in for a co-running collective kernel
 * in tools/dp_corunner_probe.py.  mpnn_debug_noop: one wave that returns at once -- the launch floor in bench.py. */
int mpnn_debug_spin(int wgs, int threads, float us, void *stream);
int mpnn_debug_noop(void *stream);

const char *mpnn_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MPNN_HIP_H */
