// Any-WIDTH forms of the exit path: mpnn_lin_fwd_gen / mpnn_lin_bwd_gen / mpnn_exit_tail_fwd_gen /
// mpnn_exit_tail_bwd_gen / mpnn_exit_ev_gen.
//
// The reference's LinTrans takes any n_chan (scripts/lib/layer_types.py:39-53) and arch_and_hypers.router builds its MLP
// from `router_n_chan` (arch_and_hypers.py:14,45-49); the tuned exit kernels (lin.hip, exit_tail.hip, exit_ev.hip) hold
// at most 16 classes and two EQUAL hidden layers of at most 16 units in registers / MFMA tiles, which is what every
// shipped spec uses.  These kernels take the SAME argument records (with R2 = width of the second hidden layer) for any
// n_cls <= 1024, R, R2 <= 256, any batch size, in training (batch statistics) and evaluation (moving averages) mode.
// They are plain: a thread per output element, loops over the contraction, every intermediate recomputed from what the
// forward pass stored (h1, h2) instead of staged -- no tile limits, no scratch, and not latency-tuned (a net that needs
// them pays ~0.2 ms per step for its exits).  The engine switches a net to them when one of its exits is outside the
// tuned kernels' limits (lib/_plan.py: Engine.generic_exits).
#include "common.h"

#define GEN_C 256            // channels of the exit's input map (coefficient table)
#define GEN_R 256            // router widths
#define GEN_K 4096           // features of a row held in LDS by the forward affine map

// act(x) of mpnn_act for one element, coefficients cA[c] = (m, gamma * rstd, beta)
__device__ __forceinline__ float gen_act(const mpnn_act &a, const float *cA, float x, int c) {
    if (a.mode == MPNN_ACT_IDENTITY) return x;
    return fmaxf((x - cA[c * 3]) * cA[c * 3 + 1] + cA[c * 3 + 2], 0.f);
}
__device__ __forceinline__ void gen_table(const mpnn_act &a, float *cA) {
    if (a.mode != MPNN_ACT_IDENTITY)
        for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
            const BnC k = bn_coef(a, c);
            cA[c * 3] = k.m; cA[c * 3 + 1] = k.gamma * k.rstd; cA[c * 3 + 2] = k.beta;
        }
    __syncthreads();
}

// ---------------------------------------------------------------------------
// y[s] = act(x) @ w[s] + b[s] (+ alpha * k_cpt * w[s][K] with extra_col[s]); s = head, router first map
// grid (row blocks of 4, records)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lin_fwd_gen_k(const mpnn_lin_fwd_args *__restrict__ tab) {
    const mpnn_lin_fwd_args &a = tab[blockIdx.y];
    const int r0 = blockIdx.x * 4;
    if (r0 >= a.n) return;
    __shared__ float cA[GEN_C * 3];
    __shared__ float xs[4 * GEN_K];
    const int K = a.HW * a.a.C, C = a.a.C, tid = threadIdx.x;
    gen_table(a.a, cA);
    const int rows = a.n - r0 < 4 ? a.n - r0 : 4;
    for (int e = tid; e < rows * K; e += 256) {
        const int r = e / K, k = e - r * K;
        xs[r * K + k] = gen_act(a.a, cA, a.a.x[(size_t)(r0 + r) * K + k], k % C);
    }
    __syncthreads();
    const int M0 = a.w[0] ? a.M[0] : 0, M1 = a.w[1] ? a.M[1] : 0, Mt = M0 + M1;
    for (int e = tid; e < rows * Mt; e += 256) {
        const int r = e / Mt, c = e - r * Mt, s = c < M0 ? 0 : 1, cc = s ? c - M0 : c, M = s ? M1 : M0;
        const float *w = a.w[s], *x = xs + r * K;
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc += x[k] * w[(size_t)k * M + cc];
        acc += a.b[s][cc];
        if (a.extra_col[s]) acc += a.alpha_cpt * a.k_cpt[r0 + r] * w[(size_t)K * M + cc];
        a.y[s][(size_t)(r0 + r) * M + cc] = acc;
    }
}

// dW[s][k][c] = sum_r act(x)[r][k] dy[s][r][c] (k = K: the k_cpt row, k = K + 1: db).  grid (feature blocks of 8, records)
__global__ __launch_bounds__(256) void lin_dw_gen_k(const mpnn_lin_bwd_args *__restrict__ tab) {
    const mpnn_lin_bwd_args &a = tab[blockIdx.y];
    const int K = a.HW * a.a.C, C = a.a.C, tid = threadIdx.x;
    const int k0 = blockIdx.x * 8;
    if (k0 > K + 1) return;
    __shared__ float cA[GEN_C * 3];
    gen_table(a.a, cA);
    const int M0 = a.w[0] ? a.M[0] : 0, M1 = a.w[1] ? a.M[1] : 0, Mt = M0 + M1;
    for (int e = tid; e < 8 * Mt; e += 256) {
        const int kf = k0 + e / Mt, c = e % Mt, s = c < M0 ? 0 : 1, cc = s ? c - M0 : c, M = s ? M1 : M0;
        if (kf > K + 1 || (kf == K && !a.extra_col[s])) continue;
        const float *dy = a.dy[s];
        float acc = 0.f;
        for (int r = 0; r < a.n; ++r) {
            const float x = kf < K ? gen_act(a.a, cA, a.a.x[(size_t)r * K + kf], kf % C)
                                   : (kf == K ? a.alpha_cpt * a.k_cpt[r] : 1.f);
            acc += x * dy[(size_t)r * M + cc];
        }
        if (kf <= K) a.dw[s][(size_t)kf * M + cc] = acc;
        else a.db[s][cc] = acc;
    }
}

// dx[r][k] = sum_s sum_c dy[s][r][c] w[s][k][c]  (gradient w.r.t. the ACTIVATED input; the consumer masks it)
__global__ __launch_bounds__(256) void lin_dx_gen_k(const mpnn_lin_bwd_args *__restrict__ tab) {
    const mpnn_lin_bwd_args &a = tab[blockIdx.y];
    if (!a.dx) return;
    const int K = a.HW * a.a.C;
    const long total = (long)a.n * K;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int r = (int)(e / K), k = (int)(e - (long)r * K);
        float acc = 0.f;
        for (int s = 0; s < 2; ++s) {
            if (!a.w[s]) continue;
            const int M = a.M[s];
            const float *dy = a.dy[s] + (size_t)r * M, *w = a.w[s] + (size_t)k * M;
            for (int c = 0; c < M; ++c) acc += dy[c] * w[c];
        }
        a.dx[e] = acc;
    }
}

// ---------------------------------------------------------------------------
// router tail + head, training or evaluation statistics.  One workgroup per exit; every cross-sample quantity (the
// BatchNorm statistics) is a loop of ONE thread per channel over the rows, every per-sample quantity a loop of one
// thread per row over the channels, with the intermediates recomputed from h1 (the previous launch's output).
// ---------------------------------------------------------------------------
struct RouterStat { float m1[GEN_R], s1[GEN_R], m2[GEN_R], s2[GEN_R]; };      // mean, rstd of both BatchNorms

__device__ __forceinline__ float gen_a1(const mpnn_exit_tail_args &a, const RouterStat &st, int r, int c) {
    return fmaxf(a.g1[c] * (a.h1[(size_t)r * a.R + c] - st.m1[c]) * st.s1[c] + a.b1[c], 0.f);
}
__device__ __forceinline__ float gen_h2(const mpnn_exit_tail_args &a, const RouterStat &st, int r, int j, int R2) {
    float h = a.bias2[j];
    for (int c = 0; c < a.R; ++c) h += gen_a1(a, st, r, c) * a.w2[c * R2 + j];
    return h;
}
// mean / rstd of column c of `col(r)` over the n rows (two passes, biased variance) + the moving averages
template <class F>
__device__ __forceinline__ void gen_bn_col(F col, int n, const mpnn_exit_tail_args &a, float *m_avg, float *v_avg, int c,
                                           float &mean, float &rstd) {
    if (a.mode != MPNN_ACT_BN_BATCH) { mean = m_avg[c]; rstd = rsqrtf(v_avg[c] + a.bn_eps); return; }
    double s = 0.0;
    for (int r = 0; r < n; ++r) s += (double)col(r);
    const float mu = (float)(s / n);
    double v = 0.0;
    for (int r = 0; r < n; ++r) { const double d = (double)col(r) - (double)mu; v += d * d; }
    const float var = (float)(v / n);
    mean = mu; rstd = rsqrtf(var + a.bn_eps);
    m_avg[c] = a.bn_decay * m_avg[c] + (1.f - a.bn_decay) * mu;
    v_avg[c] = a.bn_decay * v_avg[c] + (1.f - a.bn_decay) * var;
}

__device__ __forceinline__ void gen_head_fwd(const mpnn_exit_tail_args &a, int r) {
    const int nc = a.n_cls;
    const float *z = a.z + (size_t)r * nc, *y = a.y + (size_t)r * nc;
    float mx = z[0];
    for (int k = 1; k < nc; ++k) mx = fmaxf(mx, z[k]);
    float sum = 0.f;
    for (int k = 0; k < nc; ++k) sum += expf(z[k] - mx);
    const float inv = 1.f / sum;
    float ce = 0.f, pmax = 0.f, ymax = 0.f; int ap = 0, ay = 0;
    for (int k = 0; k < nc; ++k) {
        const float pk = expf(z[k] - mx) * inv, yk = y[k];
        ce -= yk * logf(a.eps_ce / (float)nc + (1.f - a.eps_ce) * pk);
        if (k == 0 || pk > pmax) { pmax = pk; ap = k; }
        if (k == 0 || yk > ymax) { ymax = yk; ay = k; }
    }
    a.c_err[r] = ce;
    a.d_cor[r] = ap == ay ? 1.f : 0.f;
}

__global__ __launch_bounds__(256) void exit_tail_fwd_gen_k(const mpnn_exit_tail_args *__restrict__ tab, int n_rec0) {
    const mpnn_exit_tail_args &a = tab[blockIdx.x];
    const int n = a.n, tid = threadIdx.x;
    __shared__ RouterStat st;
    if (blockIdx.x == 0 && n_rec0) {            // (the accumulators mpnn_route adds to: see mpnn_exit_tail_args)
        for (int i = tid; i < a.n_clear_f; i += 256) a.clear_f[i] = 0.f;
        for (int i = tid; i < a.n_clear_d; i += 256) a.clear_d[i] = 0.0;
        if (a.hyp_src && tid < MPNN_HYP_N) a.hyp_dst[tid] = a.hyp_src[tid];
    }
    if (a.z) for (int r = tid; r < n; r += 256) gen_head_fwd(a, r);
    if (!a.h1) return;
    const int R = a.R, R2 = a.R2 > 0 ? a.R2 : a.R, S = a.n_sinks;
    for (int c = tid; c < R; c += 256)
        gen_bn_col([&](int r) { return a.h1[(size_t)r * R + c]; }, n, a, a.m1, a.v1, c, st.m1[c], st.s1[c]);
    __syncthreads();
    for (int j = tid; j < R2; j += 256)
        gen_bn_col([&](int r) { return gen_h2(a, st, r, j, R2); }, n, a, a.m2, a.v2, j, st.m2[j], st.s2[j]);
    __syncthreads();
    if (a.bn_save)
        for (int c = tid; c < 2 * R + 2 * R2; c += 256)
            a.bn_save[c] = c < R ? st.m1[c] : c < 2 * R ? st.s1[c - R] : c < 2 * R + R2 ? st.m2[c - 2 * R] : st.s2[c - 2 * R - R2];
    for (int r = tid; r < n; r += 256) {
        float out[MPNN_MAX_SINKS];
        for (int s = 0; s < MPNN_MAX_SINKS; ++s) out[s] = s < S ? a.bias3[s] : 0.f;
        for (int j = 0; j < R2; ++j) {
            const float h = gen_h2(a, st, r, j, R2);
            if (a.h2) a.h2[(size_t)r * R2 + j] = h;
            const float a2 = fmaxf(a.g2[j] * (h - st.m2[j]) * st.s2[j] + a.b2[j], 0.f);
            for (int s = 0; s < MPNN_MAX_SINKS; ++s) if (s < S) out[s] += a2 * a.w3[j * S + s];
        }
        for (int s = 0; s < S; ++s) a.r[(size_t)r * a.r_stride + s] = out[s];
    }
}

// backward of the same (training mode): see the derivation in exit_tail.hip; every intermediate is recomputed from
// h1 / h2 / bn_save / dr.  dy2(r, j) = [a2 > 0] sum_s dr[r][s] w3[j][s]; dh2 = g2 rstd2 (dy2 - mean(dy2) - xhat2 mean(dy2 xhat2)); ...
struct BwdStat { float m1[GEN_R], s1[GEN_R], m2[GEN_R], s2[GEN_R], p2[GEN_R], q2[GEN_R], p1[GEN_R], q1[GEN_R]; };

__device__ __forceinline__ float gen_dy2(const mpnn_exit_tail_bwd_args &b, const BwdStat &st, int r, int j, int R2, float &xh) {
    const mpnn_exit_tail_args &a = b.f;
    const float h = a.h2[(size_t)r * R2 + j];
    xh = (h - st.m2[j]) * st.s2[j];
    if (a.g2[j] * xh + a.b2[j] <= 0.f) return 0.f;
    float d = 0.f;
    for (int s = 0; s < a.n_sinks; ++s) d += b.dr[(size_t)r * a.r_stride + s] * a.w3[j * a.n_sinks + s];
    return d;
}
__device__ __forceinline__ float gen_dh2(const mpnn_exit_tail_bwd_args &b, const BwdStat &st, int r, int j, int R2) {
    float xh;
    const float dy = gen_dy2(b, st, r, j, R2, xh);
    return b.f.g2[j] * st.s2[j] * (dy - st.p2[j] - xh * st.q2[j]);
}
__device__ __forceinline__ float gen_dy1(const mpnn_exit_tail_bwd_args &b, const BwdStat &st, int r, int c, int R2, float &xh) {
    const mpnn_exit_tail_args &a = b.f;
    xh = (a.h1[(size_t)r * a.R + c] - st.m1[c]) * st.s1[c];
    if (a.g1[c] * xh + a.b1[c] <= 0.f) return 0.f;
    float d = 0.f;
    for (int j = 0; j < R2; ++j) d += gen_dh2(b, st, r, j, R2) * a.w2[c * R2 + j];
    return d;
}

__global__ __launch_bounds__(256) void exit_tail_bwd_gen_k(const mpnn_exit_tail_bwd_args *__restrict__ tab) {
    const mpnn_exit_tail_bwd_args &b = tab[blockIdx.x];
    const mpnn_exit_tail_args &a = b.f;
    const int n = a.n, tid = threadIdx.x;
    if (a.z && b.dz) {                          // head: dz = w_cerr * dCE/dz
        const int nc = a.n_cls;
        for (int r = tid; r < n; r += 256) {
            const float *z = a.z + (size_t)r * nc, *y = a.y + (size_t)r * nc;
            float mx = z[0];
            for (int k = 1; k < nc; ++k) mx = fmaxf(mx, z[k]);
            float sum = 0.f;
            for (int k = 0; k < nc; ++k) sum += expf(z[k] - mx);
            const float inv = 1.f / sum, hw = b.w_cerr[r];
            float dot = 0.f;
            for (int k = 0; k < nc; ++k) {
                const float p = expf(z[k] - mx) * inv, q = a.eps_ce / (float)nc + (1.f - a.eps_ce) * p;
                dot += -hw * y[k] * (1.f - a.eps_ce) / q * p;
            }
            for (int k = 0; k < nc; ++k) {
                const float p = expf(z[k] - mx) * inv, q = a.eps_ce / (float)nc + (1.f - a.eps_ce) * p;
                b.dz[(size_t)r * nc + k] = p * (-hw * y[k] * (1.f - a.eps_ce) / q - dot);
            }
        }
    }
    if (!a.h1) return;
    const int R = a.R, R2 = a.R2 > 0 ? a.R2 : a.R, S = a.n_sinks;
    __shared__ BwdStat st;
    for (int c = tid; c < 2 * R + 2 * R2; c += 256) {
        const float v = a.bn_save[c];
        if (c < R) st.m1[c] = v; else if (c < 2 * R) st.s1[c - R] = v; else if (c < 2 * R + R2) st.m2[c - 2 * R] = v; else st.s2[c - 2 * R - R2] = v;
    }
    __syncthreads();
    const float inv = 1.f / (float)n;
    for (int s = tid; s < S; s += 256) {        // dbias3
        float acc = 0.f;
        for (int r = 0; r < n; ++r) acc += b.dr[(size_t)r * a.r_stride + s];
        b.dbias3[s] = acc;
    }
    for (int j = tid; j < R2; j += 256) {       // column j of the second BatchNorm: dW3 row, dbeta2, dgamma2, the two means
        float sdy = 0.f, sdyx = 0.f, w3g[MPNN_MAX_SINKS] = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < n; ++r) {
            float xh;
            const float dy = gen_dy2(b, st, r, j, R2, xh);
            sdy += dy; sdyx += dy * xh;
            const float a2 = fmaxf(a.g2[j] * xh + a.b2[j], 0.f);
            for (int s = 0; s < MPNN_MAX_SINKS; ++s) if (s < S) w3g[s] += a2 * b.dr[(size_t)r * a.r_stride + s];
        }
        b.db2[j] = sdy; b.dg2[j] = sdyx;
        st.p2[j] = sdy * inv; st.q2[j] = sdyx * inv;
        for (int s = 0; s < S; ++s) b.dw3[j * S + s] = w3g[s];
    }
    __syncthreads();
    for (int j = tid; j < R2; j += 256) {       // dbias2
        float acc = 0.f;
        for (int r = 0; r < n; ++r) acc += gen_dh2(b, st, r, j, R2);
        b.dbias2[j] = acc;
    }
    for (int e = tid; e < R * R2; e += 256) {   // dW2[c][j] = sum_r a1 dh2
        const int c = e / R2, j = e - c * R2;
        float acc = 0.f;
        for (int r = 0; r < n; ++r) {
            const float xh = (a.h1[(size_t)r * R + c] - st.m1[c]) * st.s1[c];
            acc += fmaxf(a.g1[c] * xh + a.b1[c], 0.f) * gen_dh2(b, st, r, j, R2);
        }
        b.dw2[e] = acc;
    }
    for (int c = tid; c < R; c += 256) {        // column c of the first BatchNorm
        float sdy = 0.f, sdyx = 0.f;
        for (int r = 0; r < n; ++r) {
            float xh;
            const float dy = gen_dy1(b, st, r, c, R2, xh);
            sdy += dy; sdyx += dy * xh;
        }
        b.db1[c] = sdy; b.dg1[c] = sdyx;
        st.p1[c] = sdy * inv; st.q1[c] = sdyx * inv;
    }
    __syncthreads();
    for (int e = tid; e < n * R; e += 256) {    // dh1
        const int r = e / R, c = e - r * R;
        float xh;
        const float dy = gen_dy1(b, st, r, c, R2, xh);
        b.dh1[e] = a.g1[c] * st.s1[c] * (dy - st.p1[c] - xh * st.q1[c]);
    }
}

// ---------------------------------------------------------------------------
// evaluation: 8 samples per workgroup end to end (see mpnn_exit_ev)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void exit_ev_gen_k(const mpnn_exit_ev_args *__restrict__ tab) {
    const mpnn_exit_ev_args &a = tab[blockIdx.y];
    int n = a.n;
    if (a.cnt) { const int c = *a.cnt; n = c < n ? c : n; }
    const int s0 = blockIdx.x * 8;
    if (s0 >= n) return;
    __shared__ float cA[GEN_C * 3];
    __shared__ float zs[8 * 1024], hs[8 * GEN_R], a1s[8 * GEN_R], a2s[8 * GEN_R];
    __shared__ int img_s[8];
    const int tid = threadIdx.x, K = a.HW * a.a.C, C = a.a.C;
    const int nc = a.w_head ? a.n_cls : 0, R = a.w1 ? a.R : 0, R2 = a.w1 ? (a.R2 > 0 ? a.R2 : a.R) : 0, S = a.w1 ? a.n_sinks : 0;
    gen_table(a.a, cA);
    if (tid < 8) img_s[tid] = s0 + tid < n ? (a.idx ? a.idx[s0 + tid] : s0 + tid) : -1;
    __syncthreads();
    const int Mt = nc + R;
    for (int e = tid; e < 8 * Mt; e += 256) {
        const int sm = e / Mt, c = e - sm * Mt, img = img_s[sm];
        if (img < 0) continue;
        const float *x = a.a.x + (size_t)img * K;
        const bool head = c < nc;
        const int cc = head ? c : c - nc, M = head ? nc : R;
        const float *w = head ? a.w_head : a.w1;
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc += gen_act(a.a, cA, x[k], k % C) * w[(size_t)k * M + cc];
        if (head) zs[sm * 1024 + cc] = acc + a.b_head[cc];
        else {
            acc += a.b1[cc];
            if (a.extra_col) acc += a.alpha_cpt * a.k_cpt[img] * a.w1[(size_t)K * R + cc];
            hs[sm * GEN_R + cc] = acc;
        }
    }
    __syncthreads();
    for (int e = tid; e < 8 * R; e += 256) {
        const int sm = e / R, c = e - sm * R;
        a1s[sm * GEN_R + c] = fmaxf(a.g1[c] * (hs[sm * GEN_R + c] - a.m1[c]) * rsqrtf(a.v1[c] + a.bn_eps) + a.be1[c], 0.f);
    }
    __syncthreads();
    for (int e = tid; e < 8 * R2; e += 256) {
        const int sm = e / R2, j = e - sm * R2;
        float h = a.bias2[j];
        for (int c = 0; c < R; ++c) h += a1s[sm * GEN_R + c] * a.w2[c * R2 + j];
        a2s[sm * GEN_R + j] = fmaxf(a.g2[j] * (h - a.m2[j]) * rsqrtf(a.v2[j] + a.bn_eps) + a.be2[j], 0.f);
    }
    __syncthreads();
    if (tid >= 8 || img_s[tid] < 0) return;
    const int img = img_s[tid];
    if (nc) {
        const float *z = zs + tid * 1024, *y = a.y + (size_t)img * nc;
        float mx = z[0];
        for (int k = 1; k < nc; ++k) mx = fmaxf(mx, z[k]);
        float sum = 0.f;
        for (int k = 0; k < nc; ++k) sum += expf(z[k] - mx);
        const float inv = 1.f / sum;
        float ce = 0.f, pmax = 0.f, ymax = 0.f; int ap = 0, ay = 0;
        for (int k = 0; k < nc; ++k) {
            const float pk = expf(z[k] - mx) * inv, yk = y[k];
            ce -= yk * logf(a.eps_ce / (float)nc + (1.f - a.eps_ce) * pk);
            if (k == 0 || pk > pmax) { pmax = pk; ap = k; }
            if (k == 0 || yk > ymax) { ymax = yk; ay = k; }
        }
        a.c_err[img] = ce;
        a.d_cor[img] = ap == ay ? 1.f : 0.f;
    }
    if (S) {
        int arg = 0; float rmax = 0.f;
        for (int s = 0; s < S; ++s) {
            float r = a.bias3[s];
            for (int j = 0; j < R2; ++j) r += a2s[tid * GEN_R + j] * a.w3[j * S + s];
            a.r[(size_t)img * a.r_stride + s] = r;
            if (s == 0 || r > rmax) { rmax = r; arg = s; }             // first index on ties (tf.argmax)
        }
        if (a.child_idx[arg]) {
            const int pos = atomicAdd(a.child_cnt[arg], 1);
            if (pos < a.n) a.child_idx[arg][pos] = img;
        }
    }
}

extern "C" int mpnn_lin_fwd_gen(const mpnn_lin_fwd_args *dev_table, int count, int n_max, void *stream) {
    if (count <= 0 || n_max <= 0) return 0;
    if (!dev_table) return MPNN_E_ARG;
    hipLaunchKernelGGL(lin_fwd_gen_k, dim3((n_max + 3) / 4, count), dim3(256), 0, (hipStream_t)stream, dev_table);
    MPNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mpnn_lin_bwd_gen(const mpnn_lin_bwd_args *dev_table, int count, int n_max, int k_max, void *stream) {
    if (count <= 0 || n_max <= 0) return 0;
    if (!dev_table || k_max <= 0) return MPNN_E_ARG;
    hipLaunchKernelGGL(lin_dw_gen_k, dim3((k_max + 2 + 7) / 8, count), dim3(256), 0, (hipStream_t)stream, dev_table);
    MPNN_LAUNCH_CHECK();
    long blocks = ((long)n_max * k_max + 1023) / 1024;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(lin_dx_gen_k, dim3((unsigned)blocks, count), dim3(256), 0, (hipStream_t)stream, dev_table);
    MPNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mpnn_exit_tail_fwd_gen(const mpnn_exit_tail_args *dev_table, int count, int n_max, void *stream) {
    if (count <= 0 || n_max <= 0) return 0;
    if (!dev_table) return MPNN_E_ARG;
    hipLaunchKernelGGL(exit_tail_fwd_gen_k, dim3(count), dim3(256), 0, (hipStream_t)stream, dev_table, 1);
    MPNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mpnn_exit_tail_bwd_gen(const mpnn_exit_tail_bwd_args *dev_table, int count, int n_max, void *stream) {
    if (count <= 0 || n_max <= 0) return 0;
    if (!dev_table) return MPNN_E_ARG;
    hipLaunchKernelGGL(exit_tail_bwd_gen_k, dim3(count), dim3(256), 0, (hipStream_t)stream, dev_table);
    MPNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mpnn_exit_ev_gen(const mpnn_exit_ev_args *dev_table, int count, int n_max, void *stream) {
    if (count <= 0 || n_max <= 0) return 0;
    if (!dev_table) return MPNN_E_ARG;
    hipLaunchKernelGGL(exit_ev_gen_k, dim3((n_max + 7) / 8, count), dim3(256), 0, (hipStream_t)stream, dev_table);
    MPNN_LAUNCH_CHECK();
    return 0;
}

// host-side limits of the any-width forms (records live in device memory: the caller validates before uploading)
extern "C" int mpnn_exit_gen_check(int C, int K, int n_cls, int R, int R2, int n_sinks) {
    if (C < 1 || C > GEN_C || K < 1 || K > GEN_K || (K % C)) return MPNN_E_SHAPE;
    if (n_cls < 0 || n_cls > 1024 || R < 0 || R > GEN_R || R2 < 0 || R2 > GEN_R) return MPNN_E_SHAPE;
    if (R && (n_sinks < 2 || n_sinks > MPNN_MAX_SINKS)) return MPNN_E_SHAPE;
    return 0;
}
