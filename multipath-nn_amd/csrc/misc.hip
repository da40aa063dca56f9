// Small HBM-bound kernels around the conv path: weight packing, BatchNorm
// backward pieces, moving averages, TALR+momentum update, branch compaction.
#include "common.h"
#include "opt_body.h"

// ---------------------------------------------------------------------------
// mpnn_pack_weights
// fwd pack [9][nchF][4][Cout][4]:  (tap, ch, gb, co, j) <- W[tap][ch*16+4gb+j][co]
// bwd pack [9][nchB][4][Cin ][4]:  (tap, ch, gb, ci, j) <- W[8-tap][ci][ch*16+4gb+j]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_k(const float *__restrict__ params, float *__restrict__ packs,
                                              const int *__restrict__ desc, f32x4 *__restrict__ zero, long zero_vec) {
    // mpnn_step_begin: the same launch clears the step's accumulators (statistics slots, loss, gradients)
    if (zero) {
        const long nb = (long)gridDim.x * gridDim.y * gridDim.z;
        const long b = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        for (long i = b * 256 + threadIdx.x; i < zero_vec; i += nb * 256) zero[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (!desc) return;                                 // (clear only: the packs are kept current by the optimizer)
    const int *d = desc + blockIdx.x * 6;
    const int src = d[0], fwd = d[1], bwd = d[2], Cin = d[3], Cout = d[4];
    const int tap = blockIdx.y;
    const float *W = params + src;
    const int t0 = blockIdx.z * 256 + threadIdx.x, tstride = gridDim.z * 256;
    if (fwd >= 0) {
        const int nch = (Cin + 15) >> 4, per_tap = nch * 16 * Cout;
        float *dst = packs + fwd + (size_t)tap * per_tap;
        // (channel counts are powers of two in every shipped spec: shifts instead of integer divisions,
        // which made this kernel VALU-bound)
        const bool p2 = (Cout & (Cout - 1)) == 0;
        const int sh = 31 - __clz(Cout);
        for (int i = t0; i < per_tap; i += tstride) {
            const int j = i & 3, q = i >> 2;
            const int co = p2 ? (q & (Cout - 1)) : q % Cout, rest = p2 ? (q >> sh) : q / Cout;   // rest = ch*4 + gb
            const int c = rest * 4 + j;
            dst[i] = c < Cin ? W[((size_t)tap * Cin + c) * Cout + co] : 0.f;
        }
    }
    if (bwd >= 0) {
        const int nch = (Cout + 15) >> 4, per_tap = nch * 16 * Cin;
        float *dst = packs + bwd + (size_t)tap * per_tap;
        const bool p2 = (Cin & (Cin - 1)) == 0;
        const int sh = 31 - __clz(Cin);
        for (int i = t0; i < per_tap; i += tstride) {
            const int j = i & 3, q = i >> 2;
            const int ci = p2 ? (q & (Cin - 1)) : q % Cin, rest = p2 ? (q >> sh) : q / Cin;
            const int co = rest * 4 + j;
            dst[i] = co < Cout ? W[((size_t)(8 - tap) * Cin + ci) * Cout + co] : 0.f;
        }
    }
}

extern "C" int mpnn_pack_weights(const float *params, float *packs, const int *desc, int n_desc, void *stream) {
    if (n_desc <= 0) return 0;
    hipLaunchKernelGGL(pack_k, dim3(n_desc, 9, 8), dim3(256), 0, (hipStream_t)stream, params, packs, desc,
                       (f32x4 *)nullptr, 0L);
    MPNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mpnn_step_begin(const float *params, float *packs, const int *desc, int n_desc,
                               void *zero, long zero_bytes, void *stream) {
    if (n_desc < 0 || (n_desc == 0 && !zero)) return MPNN_E_ARG;
    if (zero && (((size_t)zero & 15) || (zero_bytes & 15) || zero_bytes < 0)) return MPNN_E_ARG;
    if (n_desc == 0) {                                 // clear only
        long wgs = (zero_bytes / 16 + 1023) / 1024;
        wgs = wgs < 1 ? 1 : (wgs > 2048 ? 2048 : wgs);
        hipLaunchKernelGGL(pack_k, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, params, packs, (const int *)nullptr,
                           (f32x4 *)zero, zero_bytes / 16);
        MPNN_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(pack_k, dim3(n_desc, 9, 8), dim3(256), 0, (hipStream_t)stream, params, packs, desc,
                       (f32x4 *)zero, zero ? zero_bytes / 16 : 0L);
    MPNN_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// mpnn_bn_bwd_reduce / mpnn_bn_bwd_apply (elementwise forms; the conv
// epilogues hold the fused forms).  C % 4 == 0 and (C/4) | 256.
// ---------------------------------------------------------------------------
struct BnBwdP { const float *dy; const float *s; mpnn_act bn; const double *red; float *dz; double *red_out; long n_pix; int red_nslot; };

// Per-workgroup coefficient table in LDS: [C][6] = m, rstd, gamma*rstd, beta, red0/cnt, red1/cnt.
__device__ __forceinline__ void bn_table(const BnBwdP &p, float *tab) {
    const int C = p.bn.C;
    const double inv = 1.0 / (double)p.bn.cnt;
    for (int c = threadIdx.x; c < C; c += 256) {
        const BnC k = bn_coef(p.bn, c);
        float *e = tab + c * 6;
        e[0] = k.m; e[1] = k.rstd; e[2] = k.gamma * k.rstd; e[3] = k.beta;
        e[4] = p.red ? (float)(slot_sum(p.red, 2 * C, c, p.red_nslot) * inv) : 0.f;
        e[5] = p.red ? (float)(slot_sum(p.red, 2 * C, C + c, p.red_nslot) * inv) : 0.f;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_k(const BnBwdP p) {
    __shared__ float tab[1024 * 6 / 4];           // C <= 256 here (host checks)
    __shared__ double sh[256 * 8];
    bn_table(p, tab);
    const int C = p.bn.C, Q = C >> 2;             // channel quads
    const int q = threadIdx.x % Q, lane_pix = threadIdx.x / Q, ppb = 256 / Q;
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    for (long pix = (long)blockIdx.x * ppb + lane_pix; pix < p.n_pix; pix += (long)gridDim.x * ppb) {
        const size_t idx = (size_t)pix * C + q * 4;
        const f32x4 dy = *(const f32x4 *)(p.dy + idx), s = *(const f32x4 *)(p.s + idx);
        f32x4 dz;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float *e = tab + (q * 4 + j) * 6;
            const float d = s[j] - e[0];
            const float yv = d * e[2] + e[3];
            dz[j] = yv > 0.f ? dy[j] : 0.f;
            s1[j] += dz[j]; s2[j] += dz[j] * (d * e[1]);
        }
        *(f32x4 *)(p.dz + idx) = dz;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { sh[threadIdx.x * 8 + j] = s1[j]; sh[threadIdx.x * 8 + 4 + j] = s2[j]; }
    __syncthreads();
    if (threadIdx.x < C) {
        const int c = threadIdx.x, cq = c >> 2, cj = c & 3;
        double a1 = 0, a2 = 0;
        for (int r = 0; r < ppb; ++r) { a1 += sh[(r * Q + cq) * 8 + cj]; a2 += sh[(r * Q + cq) * 8 + 4 + cj]; }
        double *slot = p.red_out + (size_t)(blockIdx.x % p.red_nslot) * 2 * C;
        atomicAdd(slot + c, a1);
        atomicAdd(slot + C + c, a2);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_k(const BnBwdP p) {
    __shared__ float tab[1024 * 6 / 4];
    bn_table(p, tab);
    const int C = p.bn.C, Q = C >> 2;
    const long total = p.n_pix * Q;
    // Q divides 256 and the grid stride, so a thread keeps one channel quad.
    const float *e0 = tab + (threadIdx.x % Q) * 24;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const size_t idx = (size_t)i * 4;
        const f32x4 dz = *(const f32x4 *)(p.dz + idx), s = *(const f32x4 *)(p.s + idx);
        f32x4 g;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float *e = e0 + j * 6;
            const float xh = (s[j] - e[0]) * e[1];
            g[j] = e[2] * (dz[j] - e[4] - xh * e[5]);
        }
        *(f32x4 *)(p.dz + idx) = g;
    }
}

// mpnn_bn_relu_fwd: y = relu(bn(x)) materialised, with the coefficients (bn_coef) and the expression
// ((x - m) * (gamma * rstd) + beta, then max 0) every consumer applies while loading.
__global__ __launch_bounds__(256) void bn_relu_fwd_k(const mpnn_act a, float *__restrict__ y, long n_pix) {
    __shared__ float tab[256 * 3];
    const int C = a.C, Q = C >> 2;
    for (int c = threadIdx.x; c < C; c += 256) {
        float m = 0.f, k = 1.f, b = 0.f;
        if (a.mode != MPNN_ACT_IDENTITY) { const BnC q = bn_coef(a, c); m = q.m; k = q.gamma * q.rstd; b = q.beta; }
        tab[c * 3] = m; tab[c * 3 + 1] = k; tab[c * 3 + 2] = b;
    }
    __syncthreads();
    const long total = n_pix * Q;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int q = (int)(i % Q);
        const f32x4 x = *(const f32x4 *)(a.x + (size_t)i * 4);
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float *cc = tab + (q * 4 + j) * 3;
            v[j] = a.mode != MPNN_ACT_IDENTITY ? fmaxf((x[j] - cc[0]) * cc[1] + cc[2], 0.f) : x[j];
        }
        *(f32x4 *)(y + (size_t)i * 4) = v;
    }
}

extern "C" int mpnn_bn_relu_fwd(const mpnn_act *a, float *y, long n_pix, void *stream) {
    if (!a || !a->x || !y) return MPNN_E_ARG;
    if (a->C <= 0 || (a->C & 3) || a->C > 256 || a->shift) return MPNN_E_SHAPE;
    if (n_pix <= 0) return 0;
    long blocks = (n_pix * (a->C >> 2) + 2047) / 2048;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(bn_relu_fwd_k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a, y, n_pix);
    MPNN_LAUNCH_CHECK();
    return 0;
}

static int bn_shape_ok(int C) { return C > 0 && (C & 3) == 0 && C <= 256 && (256 % (C >> 2)) == 0; }

extern "C" int mpnn_bn_bwd_reduce(const float *dy, const mpnn_bn_ctx *ctx, float *dz, double *red_out,
                                  long n_pix, void *stream) {
    if (!dy || !ctx || !ctx->s || !dz || !red_out) return MPNN_E_ARG;
    if (!bn_shape_ok(ctx->bn.C)) return MPNN_E_SHAPE;
    if (n_pix <= 0) return 0;
    BnBwdP p = {dy, ctx->s, ctx->bn, nullptr, dz, red_out, n_pix, ctx->red_nslot < 1 ? 1 : ctx->red_nslot};
    const int ppb = 256 / (ctx->bn.C >> 2);
    long blocks = (n_pix + (long)ppb * 16 - 1) / ((long)ppb * 16);
    blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);
    hipLaunchKernelGGL(bn_bwd_reduce_k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    MPNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mpnn_bn_bwd_apply(float *dz_inout, const mpnn_bn_ctx *ctx, long n_pix, void *stream) {
    if (!dz_inout || !ctx || !ctx->s) return MPNN_E_ARG;
    if (!bn_shape_ok(ctx->bn.C)) return MPNN_E_SHAPE;
    if (n_pix <= 0) return 0;
    BnBwdP p = {nullptr, ctx->s, ctx->bn, ctx->red, dz_inout, nullptr, n_pix, ctx->red_nslot < 1 ? 1 : ctx->red_nslot};
    long blocks = (n_pix * (ctx->bn.C >> 2) + 2047) / 2048;
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    hipLaunchKernelGGL(bn_bwd_apply_k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    MPNN_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// mpnn_bn_finalize: after a training step's backward, for every conv BatchNorm
// in one launch: moving averages (layer_types.py:233-234) from the forward sums
// and dgamma / dbeta from the backward reductions.
// table: 8 ints per BN: sum_off (doubles, also the offset of the backward
// reductions in `reds`), mavg_off, vavg_off (floats in `state`), C,
// pixels per image, gamma_goff, beta_goff (floats in `grads`; -1: skip), -.
// ---------------------------------------------------------------------------
__global__ void bn_finalize_k(double *__restrict__ sums, double *__restrict__ reds,
                              float *__restrict__ state, float *__restrict__ grads,
                              const int *__restrict__ table, float decay, int n_img, double *__restrict__ sums_keep) {
    bn_finalize_body(sums, reds, state, grads, table + blockIdx.x * 8, decay, n_img, sums_keep);
}

extern "C" int mpnn_bn_finalize(double *sums, double *reds, float *state, float *grads,
                                const int *table, int n_bn, float decay, int n_img, double *sums_keep, void *stream) {
    if (n_bn <= 0) return 0;
    hipLaunchKernelGGL(bn_finalize_k, dim3(n_bn), dim3(128), 0, (hipStream_t)stream, sums, reds, state, grads, table,
                       decay, n_img, sums_keep);
    MPNN_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// mpnn_talr_momentum_step.  seg: MPNN_SEG_INTS ints per work item (include/mpnn_hip.h):
// offset, count (<= 2048), node, is_router, l2 as float bits, w_eq offset | -1, and for conv weights
// the tensor's base, Cin, Cout and the offsets of its forward / backward packs.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void talr_momentum_k(const OptP o, const int *__restrict__ seg) {
    __shared__ float wl[2048];
    opt_seg(o, seg + blockIdx.x * MPNN_SEG_INTS, nullptr, wl);
}

extern "C" int mpnn_talr_momentum_step(float *params, float *accum, const float *grads, const int *seg, int n_seg,
                                       const float *node_stat, const float *hyp, int talr, float inv_n,
                                       float grad_scale, const float *w_eq, float *packs, void *stream) {
    if (n_seg <= 0) return 0;
    const OptP o = {params, accum, grads, node_stat, hyp, talr, inv_n, grad_scale, w_eq, packs};
    hipLaunchKernelGGL(talr_momentum_k, dim3(n_seg), dim3(256), 0, (hipStream_t)stream, o, seg);
    MPNN_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// mpnn_compact_by_branch: ordered indices of samples with p_ev > 0 and their
// count; wave64 ballot + popcount prefix, wave offsets through LDS.  One
// workgroup of 1024 threads walks the batch in strides of 1024.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void compact_k(const float *__restrict__ p_ev, int n, int *__restrict__ idx_out,
                                                  int *__restrict__ count_out) {
    __shared__ int wave_cnt[16];
    __shared__ int base_sh;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) base_sh = 0;
    __syncthreads();
    for (int start = 0; start < n; start += 1024) {
        const int i = start + tid;
        const bool on = i < n && p_ev[i] > 0.f;
        const unsigned long long mask = __ballot(on);
        const int before = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wid] = __popcll(mask);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wid; ++w) woff += wave_cnt[w];
        const int base = base_sh;
        if (on) idx_out[base + woff + before] = i;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wave_cnt[w]; base_sh = base + t; }
        __syncthreads();
    }
    if (tid == 0) *count_out = base_sh;
}

extern "C" int mpnn_compact_by_branch(const float *p_ev, int n, int *idx_out, int *count_out, void *stream) {
    if (!p_ev || !idx_out || !count_out) return MPNN_E_ARG;
    hipLaunchKernelGGL(compact_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, p_ev, n, idx_out, count_out);
    MPNN_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------
// mpnn_augment_batch: the training-batch assembly of scripts/lib/data.py:10-34 on the device.
//   out[i][u][v][c] = a_i[u + du_i][v + dv_i]  if that pixel exists, else  mean_{u,v} a_i[u][v][c]
//   a_i = x_src[j_i], mirrored along v when flip_i (data.py:10-11 rand_flip, :13-22 rand_shift);
//   y_out[i] = y_src[j_i].
// The draws (j, flip, du, dv) come from the host with the reference's RNG call sequence; the
// dataset stays resident in HBM.  One workgroup per output image: fp64 per-channel mean through
// LDS, then the gather.  C <= 4.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void augment_body(const int i, const float *__restrict__ x_src, const float *__restrict__ y_src,
                                             const int *__restrict__ draw, float *__restrict__ x_out,
                                             float *__restrict__ y_out, int H, int W, int C, int n_cls) {
    const int tid = threadIdx.x;
    const int j = draw[i * 4], flip = draw[i * 4 + 1], du = draw[i * 4 + 2], dv = draw[i * 4 + 3];
    const float *a = x_src + (size_t)j * H * W * C;
    __shared__ double part[256][4];
    __shared__ float fill[4];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int px = tid; px < H * W; px += 256)
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c < C) acc[c] += (double)a[(size_t)px * C + c];
#pragma unroll
    for (int c = 0; c < 4; ++c) part[tid][c] = acc[c];
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if (tid < w)
#pragma unroll
            for (int c = 0; c < 4; ++c) part[tid][c] += part[tid + w][c];
        __syncthreads();
    }
    if (tid < 4) fill[tid] = (float)(part[0][tid] / (double)(H * W));
    __syncthreads();
    float *o = x_out + (size_t)i * H * W * C;
    for (int e = tid; e < H * W * C; e += 256) {
        const int c = e % C, px = e / C, u = px / W, v = px - u * W;
        const int su = u + du, sv = v + dv;
        float val = fill[c];
        if ((unsigned)su < (unsigned)H && (unsigned)sv < (unsigned)W)
            val = a[((size_t)su * W + (flip ? W - 1 - sv : sv)) * C + c];
        o[e] = val;
    }
    if (y_src && y_out)
        for (int k = tid; k < n_cls; k += 256) y_out[(size_t)i * n_cls + k] = y_src[(size_t)j * n_cls + k];
}

__global__ __launch_bounds__(256) void augment_k(const float *__restrict__ x_src, const float *__restrict__ y_src,
                                                 const int *__restrict__ draw, float *__restrict__ x_out,
                                                 float *__restrict__ y_out, int H, int W, int C, int n_cls) {
    augment_body(blockIdx.x, x_src, y_src, draw, x_out, y_out, H, W, C, n_cls);
}

// the batches of `count` consumers (co-trained nets, lib/_co.py) from one dataset in one launch: consumer r's record
// buffer and destinations in dst[r], n images each
__global__ __launch_bounds__(256) void augment_multi_k(const float *__restrict__ x_src, const float *__restrict__ y_src,
                                                       const mpnn_augment_dst *__restrict__ dst, int n, int H, int W, int C,
                                                       int n_cls) {
    const int r = blockIdx.x / n;
    const mpnn_augment_dst d = dst[r];
    augment_body(blockIdx.x - r * n, x_src, y_src, d.draw, d.x_out, d.y_out, H, W, C, n_cls);
}

extern "C" int mpnn_augment_batch_multi(const float *x_src, const float *y_src, const mpnn_augment_dst *dev_table, int count,
                                        int n, int H, int W, int C, int n_cls, void *stream) {
    if (n <= 0 || count <= 0) return 0;
    if (!x_src || !dev_table || C < 1 || C > 4 || H < 1 || W < 1) return MPNN_E_ARG;
    hipLaunchKernelGGL(augment_multi_k, dim3(n * count), dim3(256), 0, (hipStream_t)stream, x_src, y_src, dev_table, n, H, W, C, n_cls);
    MPNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mpnn_augment_batch(const float *x_src, const float *y_src, const int *draw, float *x_out, float *y_out,
                                  int n, int H, int W, int C, int n_cls, void *stream) {
    if (n <= 0) return 0;
    if (!x_src || !draw || !x_out || C < 1 || C > 4 || H < 1 || W < 1) return MPNN_E_ARG;
    hipLaunchKernelGGL(augment_k, dim3(n), dim3(256), 0, (hipStream_t)stream, x_src, y_src, draw, x_out, y_out, H, W, C, n_cls);
    MPNN_LAUNCH_CHECK();
    return 0;
}

int mpnn_reserved_cus_g = 0;

extern "C" int mpnn_set_reserved_cus(int cus) {
    const int prev = mpnn_reserved_cus_g;
    if (cus >= 0) mpnn_reserved_cus_g = cus;
    return prev;
}

__global__ void spin_k(long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ void noop_k() {}

extern "C" int mpnn_debug_spin(int wgs, int threads, float us, void *stream) {
    if (wgs <= 0) return 0;
    if (threads < 64 || threads > 1024 || us < 0.f) return MPNN_E_ARG;
    hipLaunchKernelGGL(spin_k, dim3(wgs), dim3(threads), 0, (hipStream_t)stream, (long)(us * 100.f));
    MPNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mpnn_debug_noop(void *stream) {
    hipLaunchKernelGGL(noop_k, dim3(1), dim3(64), 0, (hipStream_t)stream);
    MPNN_LAUNCH_CHECK();
    return 0;
}

int mpnn_trace_install_fwd(void *buf);
int mpnn_trace_install_dgrad(void *buf);
int mpnn_trace_install_wgrad(void *buf);
int mpnn_trace_install_lin(void *buf);
int mpnn_trace_install_tail(void *buf);
int mpnn_trace_install_route(void *buf);
int mpnn_trace_install_level(void *buf);
int mpnn_trace_install_level_small(void *buf);

extern "C" int mpnn_debug_set_trace(unsigned long long *buf) {
    int rc = mpnn_trace_install_fwd(buf);
    if (!rc) rc = mpnn_trace_install_dgrad(buf);
    if (!rc) rc = mpnn_trace_install_wgrad(buf);
    if (!rc) rc = mpnn_trace_install_level(buf);
    if (!rc) rc = mpnn_trace_install_level_small(buf);
    if (!rc) rc = mpnn_trace_install_lin(buf);
    if (!rc) rc = mpnn_trace_install_tail(buf);
    if (!rc) rc = mpnn_trace_install_route(buf);
    return rc;
}

extern "C" const char *mpnn_version(void) { return "mpnn_hip 0.1 (gfx950)"; }
