// mpnn_conv_nhwc_{fwd,dgrad,wgrad}: the single-scale `Conv` layer of the operator surface
// (scripts/lib/layer_types.py:55-74): y = b + conv2d_same(act(x), w), supp x supp filters, supp = 3 or 1.
//
//   supp 3: the 3x3 bodies of the multiscale path (conv_kernel.h) -- the entry points below fill their
//           argument records and forward: one conv is a MultiscaleConvMax scale without a vert operand.
//   supp 1: a 1x1 convolution is a plain GEMM over pixels, [n*H*W, Cin] x [Cin, Cout]: its own three
//           small MFMA kernels below (v_mfma_f32_16x16x4_f32, 64 pixels per workgroup).
//
// act(x) = the activation applied while loading (identity, ReLU -- the `Rect` that follows a `Conv`,
// layer_types.py:76-79 -- or BatchNorm+ReLU), as everywhere in this library.
#include "common.h"

// ------------------------------------------------------------------ 1x1: out = [bias +] act(A) B
// A: [M, K] rows = pixels; B(k, n) = TRANSB ? Bm[n * ldb + k] : Bm[k * ldb + n]; out: [M, N].
// Optional epilogue mask (input gradient through the producer's ReLU): out *= [mask_src > 0].
struct Gemm1P {
    mpnn_act a;  long M;  int K, N;
    const float *B;  int ldb;  const float *bias;  float *out;
    const float *mask_src;               // [M, N] pre-activation values of the producer or NULL
};

template <bool TRANSB>
__global__ __launch_bounds__(256) void gemm1x1_k(const Gemm1P p) {
    __shared__ float cA[256 * 3];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool act = p.a.mode != MPNN_ACT_IDENTITY;
    if (act) {
        for (int c = tid; c < p.K; c += 256) {
            const BnC k = bn_coef(p.a, c);
            cA[c * 3] = k.m; cA[c * 3 + 1] = k.gamma * k.rstd; cA[c * 3 + 2] = k.beta;
        }
    }
    __syncthreads();
    const int K = p.K, N = p.N;
    const bool vecA = (K & 3) == 0, vecB = TRANSB && (p.ldb & 3) == 0;
    for (long t0 = (long)blockIdx.x * 64 + wid * 16; t0 < p.M; t0 += (long)gridDim.x * 64) {
        const long row = t0 + li;
        const bool rv = row < p.M;
        const float *ar = p.a.x + (rv ? row : 0) * (long)K;
        for (int n0 = 0; n0 < N; n0 += 16) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const int col = n0 + li;
            const bool cv = col < N;
            for (int kb = 0; kb < K; kb += 16) {
                const int k = kb + 4 * g;
                float x[4], b[4];
                if (vecA && k + 3 < K) { const f32x4 v = *(const f32x4 *)(ar + k); x[0] = v[0]; x[1] = v[1]; x[2] = v[2]; x[3] = v[3]; }
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[j] = k + j < K ? ar[k + j] : 0.f;
                }
                if (vecB && k + 3 < K) {
                    const f32x4 v = *(const f32x4 *)(p.B + (size_t)(cv ? col : 0) * p.ldb + k);
                    b[0] = v[0]; b[1] = v[1]; b[2] = v[2]; b[3] = v[3];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int kk = k + j < K ? k + j : 0;
                        b[j] = TRANSB ? p.B[(size_t)(cv ? col : 0) * p.ldb + kk] : p.B[(size_t)kk * p.ldb + (cv ? col : 0)];
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float xv = x[j];
                    if (act && k + j < K) { const float *cc = cA + (k + j) * 3; xv = fmaxf((xv - cc[0]) * cc[1] + cc[2], 0.f); }
                    const bool on = rv && k + j < K;
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(on ? xv : 0.f, (cv && k + j < K) ? b[j] : 0.f, acc, 0, 0, 0);
                }
            }
            mfma_drain();
            // D: row = 4g + r (pixel of the 16-row tile), col = li
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long orow = t0 + 4 * g + r;
                if (orow < p.M && cv) {
                    float v = acc[r] + (p.bias ? p.bias[col] : 0.f);
                    if (p.mask_src) v = p.mask_src[orow * N + col] > 0.f ? v : 0.f;
                    p.out[orow * N + col] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ 1x1 weight gradient
// dW[c][co] += sum_p act(x[p][c]) g[p][co];  db[co] += sum_p g[p][co]   (ADDED into zeroed tensors:
// one fp32 atomic per element and workgroup -- a few dozen workgroups per 16x16 tile of dW).
struct Wg1P { mpnn_act a;  long M;  int Cin, Cout;  const float *g;  float *dw, *db;  int splits; };

__global__ __launch_bounds__(256) void wgrad1x1_k(const Wg1P p) {
    __shared__ float cA[256 * 3];
    __shared__ f32x4 red[2][4][64];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool act = p.a.mode != MPNN_ACT_IDENTITY;
    if (act) {
        for (int c = tid; c < p.Cin; c += 256) {
            const BnC k = bn_coef(p.a, c);
            cA[c * 3] = k.m; cA[c * 3 + 1] = k.gamma * k.rstd; cA[c * 3 + 2] = k.beta;
        }
    }
    __syncthreads();
    const int nct = (p.Cout + 15) >> 4;
    const int ci0 = (blockIdx.y / nct) * 16, co0 = (blockIdx.y % nct) * 16;
    const int ci = ci0 + li, co = co0 + li;
    const bool civ = ci < p.Cin, cov = co < p.Cout;
    float m = 0.f, k1 = 1.f, k2 = 0.f;
    if (act && civ) { m = cA[ci * 3]; k1 = cA[ci * 3 + 1]; k2 = cA[ci * 3 + 2]; }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, accb = {0.f, 0.f, 0.f, 0.f};
    const float one_hot = li == 0 ? 1.f : 0.f;
    // a wave takes 16 pixels per step: 4 MFMAs of 4 pixels, their loads issued together
    for (long p0 = ((long)blockIdx.x * 4 + wid) * 16; p0 < p.M; p0 += (long)p.splits * 64) {
        float a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long px = p0 + 4 * j + g;
            const bool ok = px < p.M;
            a[j] = (ok && civ) ? p.a.x[px * p.Cin + ci] : 0.f;
            b[j] = (ok && cov) ? p.g[px * p.Cout + co] : 0.f;
            if (act) a[j] = (ok && civ) ? fmaxf((a[j] - m) * k1 + k2, 0.f) : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
            accb = __builtin_amdgcn_mfma_f32_16x16x4f32(one_hot, b[j], accb, 0, 0, 0);
        }
    }
    mfma_drain();
    red[0][wid][lane] = acc;  red[1][wid][lane] = accb;
    __syncthreads();
    if (wid == 0) {
        const f32x4 s = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
        const f32x4 sb = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
        // D: row = 4g + r (input channel of the tile), col = li (output channel)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = ci0 + 4 * g + r;
            if (c < p.Cin && cov) atomicAdd(p.dw + (size_t)c * p.Cout + co, s[r]);
        }
        if (ci0 == 0 && g == 0 && cov && p.db) atomicAdd(p.db + co, sb[0]);
    }
}

// ------------------------------------------------------------------ entry points
static int shape1_ok(int C) { return C > 0 && C <= 256; }

extern "C" int mpnn_conv_nhwc_fwd(const mpnn_conv_nhwc_fwd_args *a, void *stream) {
    if (!a || !a->a.x || !a->w || !a->bias || !a->out) return MPNN_E_ARG;
    if (a->n <= 0) return 0;
    if (a->supp == 3) {
        mpnn_conv_fwd_args f = {};
        f.a = a->a;  f.wa_pack = a->w;  f.bias = a->bias;  f.out = a->out;
        f.out_nslot = 1;  f.n = a->n;  f.H = a->H;  f.W = a->W;  f.Cout = a->Cout;
        return mpnn_msconv_fwd(&f, stream);
    }
    if (a->supp != 1 || a->a.shift) return MPNN_E_SHAPE;
    if (!shape1_ok(a->a.C) || !shape1_ok(a->Cout)) return MPNN_E_SHAPE;
    Gemm1P p = {};
    p.a = a->a;  p.M = (long)a->n * a->H * a->W;  p.K = a->a.C;  p.N = a->Cout;
    p.B = a->w;  p.ldb = a->Cout;  p.bias = a->bias;  p.out = a->out;
    long blocks = (p.M + 63) / 64;
    blocks = blocks > 2048 ? 2048 : blocks;
    hipLaunchKernelGGL(gemm1x1_k<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    MPNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mpnn_conv_nhwc_dgrad(const mpnn_conv_nhwc_dgrad_args *a, void *stream) {
    if (!a || !a->g || !a->w || !a->dx) return MPNN_E_ARG;
    if (a->n <= 0) return 0;
    if (a->supp == 3) {
        // the producer's ReLU mask rides in the BatchNorm-backward epilogue of the 3x3 body with identity
        // coefficients (MPNN_ACT_RELU: m = 0, gamma * rstd = 1, beta = 0); its reductions go to `scratch`
        mpnn_dgrad_horz_args h = {};
        mpnn_bn_ctx ctx = {};
        h.g = a->g;  h.Cg = a->Cg;  h.w_pack = a->w;  h.out = a->dx;
        h.n = a->n;  h.H = a->H;  h.W = a->W;  h.Cout = a->Cin;
        if (a->relu_src) {
            if (!a->scratch) return MPNN_E_ARG;
            ctx.s = a->relu_src;  ctx.bn.C = a->Cin;  ctx.bn.mode = MPNN_ACT_RELU;  ctx.bn.cnt = 1;  ctx.bn.nslot = 1;
            ctx.red_nslot = 1;
            h.prev = &ctx;  h.red_out = a->scratch;
        }
        return mpnn_msconv_dgrad_horz(&h, stream);
    }
    if (a->supp != 1) return MPNN_E_SHAPE;
    if (!shape1_ok(a->Cg) || !shape1_ok(a->Cin)) return MPNN_E_SHAPE;
    Gemm1P p = {};
    p.a.x = a->g;  p.a.C = a->Cg;  p.a.mode = MPNN_ACT_IDENTITY;
    p.M = (long)a->n * a->H * a->W;  p.K = a->Cg;  p.N = a->Cin;
    p.B = a->w;  p.ldb = a->Cg;  p.out = a->dx;  p.mask_src = a->relu_src;
    long blocks = (p.M + 63) / 64;
    blocks = blocks > 2048 ? 2048 : blocks;
    hipLaunchKernelGGL(gemm1x1_k<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    MPNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mpnn_conv_nhwc_wgrad(const mpnn_conv_nhwc_wgrad_args *a, void *stream) {
    if (!a || !a->a.x || !a->g || !a->dw || !a->db) return MPNN_E_ARG;
    if (a->n <= 0) return 0;
    if (a->supp == 3) {
        mpnn_wgrad_args w = {};
        w.a = a->a;  w.g = a->g;  w.dwa = a->dw;  w.db = a->db;
        w.n = a->n;  w.H = a->H;  w.W = a->W;  w.Cout = a->Cout;
        w.n_split = a->n_split < 1 ? 1 : a->n_split;  w.split_stride = a->split_stride;
        return mpnn_msconv_wgrad(&w, stream);
    }
    if (a->supp != 1 || a->a.shift) return MPNN_E_SHAPE;
    if (!shape1_ok(a->a.C) || !shape1_ok(a->Cout)) return MPNN_E_SHAPE;
    Wg1P p = {};
    p.a = a->a;  p.M = (long)a->n * a->H * a->W;  p.Cin = a->a.C;  p.Cout = a->Cout;
    p.g = a->g;  p.dw = a->dw;  p.db = a->db;
    long splits = (p.M + 1023) / 1024;                       // >= 16 steps of 64 pixels per workgroup
    p.splits = (int)(splits < 1 ? 1 : (splits > 64 ? 64 : splits));
    const int tiles = ((p.Cin + 15) / 16) * ((p.Cout + 15) / 16);
    hipLaunchKernelGGL(wgrad1x1_k, dim3(p.splits, tiles), dim3(256), 0, (hipStream_t)stream, p);
    MPNN_LAUNCH_CHECK();
    return 0;
}
