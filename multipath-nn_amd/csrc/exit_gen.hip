// Any-WIDTH forms of the exit path: mpnn_lin_fwd_gen / mpnn_lin_bwd_gen / mpnn_exit_tail_fwd_gen /
// mpnn_exit_tail_bwd_gen / mpnn_exit_ev_gen.
//
// The reference's LinTrans takes any n_chan (scripts/lib/layer_types.py:39-53) and arch_and_hypers.router builds its MLP
// from `router_n_chan` (arch_and_hypers.py:14,45-49); the tuned exit kernels (lin.hip, exit_tail.hip, exit_ev.hip) hold
// at most 16 classes and two EQUAL hidden layers of at most 16 units in registers / MFMA tiles, which is what every
// shipped spec uses.  These kernels take the SAME argument records (with R2 = width of the second hidden layer) for any
// n_cls <= 1024, R, R2 <= 256, any batch size, in training (batch statistics) and evaluation (moving averages) mode.
// The engine switches a net to them when one of its exits is outside the tuned kernels' limits (lib/_plan.py:
// Engine.generic_exits).
//
// Round 5 rewrite.  The first version (round 4) was a thread per output element with every intermediate RECOMPUTED from
// what the forward pass stored -- a 100-class / 32-32-router net ran 8x slower per step than the shipped one (exit tails
// 0.9 + 1.5 ms: ONE workgroup per exit, 32 of its 256 threads busy in the BatchNorm column passes, each recomputing an
// R x R2 product per element; profiles/r05_wide_exits.txt).  Now:
//   * the two affine maps over the exit's K = H*W*C features (all the arithmetic there is) run on
//     v_mfma_f32_16x16x4_f32 tiles, looping the tuned kernels' 16-column tile over the head's and the router's widths:
//     forward = (16-row tile, 16-column tile) per workgroup, the four waves split K and meet in LDS; backward = a
//     16-feature tile per workgroup: dW for every column tile (contraction over the batch rows, the waves split the
//     rows) and dX for every row tile (contraction over the columns, the weight tile and the dY tiles staged in LDS);
//   * the router tail materialises what it needs once (h2 by the forward pass, dh2 in a scratch map, dy1 in the dh1
//     output) and every pass uses all 256 threads: column sums as (column, row group) partials that meet in LDS, the two
//     small products (a1' dh2, dh2 w2') from row chunks staged in LDS; the head (softmax, cross-entropy, arg-max) runs
//     in workgroups of its own, four threads per sample.
#include <atomic>
#include "common.h"

#define GEN_C 256            // channels of the exit's input map (coefficient table)
#define GEN_R 256            // router widths
#define GEN_K 65536          // features of the exit's input map (H * W * C)

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
// A value another wave of THIS workgroup stored to global memory in an earlier phase.  Plain loads are enough: the phase
// ends with phase_barrier (every store acknowledged, then the barrier), the vector L1 is write-through and starts every
// kernel invalidated, and no such buffer (h2, the dh2 scratch, the dy1 values in dh1) is read in a kernel before its
// last write by another thread -- so a line can only enter the L1 after its final contents reached the L2.
__device__ __forceinline__ float ld_l2(const float *p) { return *p; }
__device__ __forceinline__ void phase_barrier() {          // the phase's global stores have left the wave, then the barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// ---------------------------------------------------------------------------
// y[s] = act(x) @ w[s] + b[s] (+ alpha * k_cpt * w[s][K] with extra_col[s]); s = head, router first map.
// grid (16-row tiles, 16-column tiles of both sets, records); the four waves split K (LF_UN 16-feature blocks per trip, all
// their loads issued before the first MFMA), partial tiles meet in LDS.
// ---------------------------------------------------------------------------
// One (16-row, 16-column) output tile: rows = sample slots n0 .. n0 + 15 of `n`; slot -> image through idx (routed
// evaluation) or the identity; outputs, k_cpt and the input rows are indexed by IMAGE.
__device__ __forceinline__ void gen_lin_tile(const mpnn_act &act, int HW, int n, const int *__restrict__ idx, int n0,
                                             const float *__restrict__ w, const float *__restrict__ bias, int M, int c0,
                                             bool extra, float alpha, const float *__restrict__ kcpt, float *__restrict__ y,
                                             float *cA, float *red) {
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15, wid = tid >> 6;
    const int C = act.C, K = HW * C;
    gen_table(act, cA);
    const int slot = n0 + li;
    const bool valid = slot < n;
    const int img = valid ? (idx ? idx[slot] : slot) : 0;
    const float *xrow = act.x + (size_t)img * K;
    const int col = c0 + li;
    const bool cv = col < M;
    const int colc = cv ? col : 0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    constexpr int LF_UN = 4;
    const int nkb = (K + 15) >> 4;
    for (int kb = wid * LF_UN; kb < nkb; kb += 4 * LF_UN) {
        float xv[LF_UN][4], bv[LF_UN][4];
#pragma unroll
        for (int u = 0; u < LF_UN; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kk = (kb + u) * 16 + 4 * g + j;
                const int kc = kk < K ? kk : 0;
                xv[u][j] = xrow[kc];
                bv[u][j] = w[(size_t)kc * M + colc];
            }
#pragma unroll
        for (int u = 0; u < LF_UN; ++u) {
            const int k0 = (kb + u) * 16 + 4 * g;
            int c = k0 % C;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool on = k0 + j < K;
                const float xa = (on && valid) ? gen_act(act, cA, xv[u][j], c) : 0.f;
                const float xb = (on && cv) ? bv[u][j] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, xb, acc, 0, 0, 0);
                c = c + 1 < C ? c + 1 : 0;
            }
        }
    }
    mfma_drain();
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wid * 256 + lane * 4 + r] = acc[r];
    __syncthreads();
    {   // D layout: element e = lane * 4 + r -> row 4 (lane / 16) + r, column lane % 16
        const int e = tid, l = e >> 2, r = e & 3;
        const int oslot = n0 + (l >> 4) * 4 + r, ocol = c0 + (l & 15);
        if (oslot < n && ocol < M) {
            const int oimg = idx ? idx[oslot] : oslot;
            float v = bias[ocol];
            v += ((red[e] + red[256 + e]) + (red[512 + e] + red[768 + e]));
            if (extra) v += alpha * kcpt[oimg] * w[(size_t)K * M + ocol];
            y[(size_t)oimg * M + ocol] = v;
        }
    }
}

__global__ __launch_bounds__(256) void lin_fwd_gen_k(const mpnn_lin_fwd_args *__restrict__ tab) {
    const mpnn_lin_fwd_args &a = tab[blockIdx.z];
    const int n0 = blockIdx.x * 16;
    if (n0 >= a.n) return;
    const int M0 = a.w[0] ? a.M[0] : 0, M1 = a.w[1] ? a.M[1] : 0;
    const int T0 = (M0 + 15) >> 4, T1 = (M1 + 15) >> 4;
    int ct = blockIdx.y;
    if (ct >= T0 + T1) return;
    const int s = ct >= T0 ? 1 : 0;
    if (s) ct -= T0;
    __shared__ float cA[GEN_C * 3];
    __shared__ float red[4 * 256];
    gen_lin_tile(a.a, a.HW, a.n, nullptr, n0, a.w[s], a.b[s], s ? M1 : M0, ct * 16, a.extra_col[s] != 0, a.alpha_cpt, a.k_cpt,
                 a.y[s], cA, red);
}

// ---------------------------------------------------------------------------
// Backward of the same, one 16-FEATURE tile kf0 .. kf0 + 15 per workgroup (features K and K + 1 are the k_cpt row and the
// bias):   dW[s][k][c] = sum_r act(x)[r][k] dy[s][r][c]      D[i = k][j = c], inner index r: the waves split the rows
//          dx[r][k]    = sum_s sum_c dy[s][r][c] w[s][k][c]  D[i = r][j = k], inner index c (both sets' columns
//          concatenated, LB_CB at a time): the weight tile and each wave's dY row tile staged in LDS (pitch LB_CB + 1)
// dx is the gradient w.r.t. the ACTIVATED input; the consumer masks it.  grid (feature tiles, records, 2): z = 0 the weight
// gradients, z = 1 the input gradient of the tile -- two short chains side by side instead of one long one.
// ---------------------------------------------------------------------------
#define LB_CB 128
#define LB_P (LB_CB + 1)
__global__ __launch_bounds__(256) void lin_bwd_gen_k(const mpnn_lin_bwd_args *__restrict__ tab) {
    const mpnn_lin_bwd_args &a = tab[blockIdx.y];
    const int C = a.a.C, K = a.HW * C, n = a.n;
    const int kf0 = blockIdx.x * 16;
    if (kf0 > K + 1) return;
    __shared__ float cA[GEN_C * 3];
    __shared__ float wL[16 * LB_P];
    __shared__ float dyL[4][16 * LB_P];
    __shared__ float red[4 * 256];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15, wid = tid >> 6;
    gen_table(a.a, cA);
    const int M0 = a.w[0] ? a.M[0] : 0, M1 = a.w[1] ? a.M[1] : 0;
    // ---- dW, db ----
    const int kf = kf0 + li, kfc = kf < K ? kf % C : 0;
#pragma unroll 1
    for (int s = 0; s < 2 && blockIdx.z == 0; ++s) {
        const int M = s ? M1 : M0;
        if (!M) continue;
        const float *__restrict__ dy = a.dy[s];
        const bool extra = a.extra_col[s] != 0;
#pragma unroll 1
        for (int ct0 = 0; ct0 * 16 < M; ct0 += 8) {
            f32x4 acc[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int r0 = wid * 4; r0 < n; r0 += 16) {
                const int r = r0 + g;
                const bool ok = r < n;
                const int rc = ok ? r : 0;
                float xa = 0.f;
                if (kf < K) xa = gen_act(a.a, cA, a.a.x[(size_t)rc * K + kf], kfc);
                else if (kf == K) xa = extra ? a.alpha_cpt * a.k_cpt[rc] : 0.f;
                else if (kf == K + 1) xa = 1.f;
                xa = ok ? xa : 0.f;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int col = (ct0 + t) * 16 + li;
                    const float b = (ok && col < M) ? dy[(size_t)rc * M + col] : 0.f;
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, b, acc[t], 0, 0, 0);
                }
            }
            mfma_drain();
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if ((ct0 + t) * 16 >= M) break;                     // (uniform)
#pragma unroll
                for (int q = 0; q < 4; ++q) red[wid * 256 + lane * 4 + q] = acc[t][q];
                __syncthreads();
                const int e = tid, l = e >> 2, q = e & 3;
                const int kfo = kf0 + (l >> 4) * 4 + q, col = (ct0 + t) * 16 + (l & 15);
                const float v = (red[e] + red[256 + e]) + (red[512 + e] + red[768 + e]);
                if (col < M) {
                    if (kfo < K || (kfo == K && extra)) a.dw[s][(size_t)kfo * M + col] = v;
                    else if (kfo == K + 1) a.db[s][col] = v;
                }
                __syncthreads();
            }
        }
    }
    // ---- dX ----
    if (blockIdx.z == 0 || !a.dx || kf0 >= K) return;
    const int Mt = M0 + M1, ntile = (n + 15) >> 4;
#pragma unroll 1
    for (int it = 0; it * 4 < ntile; ++it) {
        const int rt = it * 4 + wid, r0 = rt * 16;
        const bool tv = rt < ntile;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int cc0 = 0; cc0 < Mt; cc0 += LB_CB) {
            const int cw = Mt - cc0 < LB_CB ? Mt - cc0 : LB_CB;
            __syncthreads();                                         // (the previous chunk's readers are done)
            {   // the weight tile [16 features][cw columns]: thread = (column, feature parity)
                const int c = tid & (LB_CB - 1), col = cc0 + c, s = col >= M0 ? 1 : 0, cc = s ? col - M0 : col, Ms = s ? M1 : M0;
                const bool cok = c < cw;
                const float *__restrict__ ws = a.w[s];
                for (int f = tid >> 7; f < 16; f += 2) {
                    const int k = kf0 + f;
                    wL[f * LB_P + c] = (cok && k < K) ? ws[(size_t)k * Ms + cc] : 0.f;
                }
                // this wave's dY tile [16 rows][cw columns]: lane = column (two per lane)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int c2 = lane + 64 * h, col2 = cc0 + c2, s2 = col2 >= M0 ? 1 : 0, cc2 = s2 ? col2 - M0 : col2, Ms2 = s2 ? M1 : M0;
                    const bool ok2 = c2 < cw && tv;
                    const float *__restrict__ d2 = a.dy[s2];
                    for (int rr = 0; rr < 16; ++rr) {
                        const int r = r0 + rr;
                        dyL[wid][rr * LB_P + c2] = (ok2 && r < n) ? d2[(size_t)r * Ms2 + cc2] : 0.f;
                    }
                }
            }
            __syncthreads();
            const int cend = (cw + 3) & ~3;
            for (int c = 0; c < cend; c += 4)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dyL[wid][li * LB_P + c + g], wL[li * LB_P + c + g], acc, 0, 0, 0);
        }
        mfma_drain();
        if (tv) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + 4 * g + q, k = kf0 + li;
                if (r < n && k < K) a.dx[(size_t)r * K + k] = acc[q];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Column sums over the batch rows with ALL threads: NV values per (row, column) from f(r, c, v), column c's totals to
// out(c, tot).  256 threads = (column, row group) pairs; the groups' fp64 partials meet in LDS in group order.
// ---------------------------------------------------------------------------
#define TT 1024              // threads of the exit-tail workgroups (16 waves: the phases are chains of dependent memory round trips)
template <int NV, class F, class O>
__device__ __forceinline__ void col_sums(int n, int W, double *lds /* [NV * TT] */, F f, O out) {
    const int tid = threadIdx.x;
    for (int cb = 0; cb < W; cb += TT) {
        const int Wb = W - cb < TT ? W - cb : TT, G = TT / Wb;
        const int grp = tid / Wb, c = cb + tid - grp * Wb;
        double acc[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) acc[k] = 0.0;
        if (grp < G)
            for (int r = grp; r < n; r += G) {
                float v[NV];
                f(r, c, v);
#pragma unroll
                for (int k = 0; k < NV; ++k) acc[k] += (double)v[k];
            }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NV; ++k) lds[k * TT + tid] = acc[k];
        __syncthreads();
        if (tid < Wb) {
            double tot[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                double t = 0.0;
                for (int gq = 0; gq < G; ++gq) t += lds[k * TT + gq * Wb + tid];
                tot[k] = t;
            }
            out(cb + tid, tot);
        }
    }
    __syncthreads();
}

struct RouterStat { float m1[GEN_R], s1[GEN_R], m2[GEN_R], s2[GEN_R], p2[GEN_R], q2[GEN_R], p1[GEN_R], q1[GEN_R]; };

// Softmax + CrossEntropyError of TT / HT samples per workgroup, HT threads per sample (classes part, part + HT, ...)
#define HT 16
__device__ __forceinline__ float quad_max(float v) {
#pragma unroll
    for (int m = 1; m < HT; m <<= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ float quad_sum(float v) {
#pragma unroll
    for (int m = 1; m < HT; m <<= 1) v += __shfl_xor(v, m);
    return v;
}

__device__ __forceinline__ void gen_head_fwd(const mpnn_exit_tail_args &a, int r0) {
    const int tid = threadIdx.x, r = r0 + tid / HT, part = tid % HT, nc = a.n_cls;
    const bool ok = r < a.n;
    const float *z = a.z + (size_t)(ok ? r : 0) * nc, *y = a.y + (size_t)(ok ? r : 0) * nc;
    float mx = -3.0e38f;
    for (int k = part; k < nc; k += HT) mx = fmaxf(mx, z[k]);
    mx = quad_max(mx);
    float sum = 0.f;
    for (int k = part; k < nc; k += HT) sum += expf(z[k] - mx);
    sum = quad_sum(sum);
    const float inv = 1.f / sum;
    float ce = 0.f, pmax = -1.f, ymax = -3.0e38f; int ap = nc, ay = nc;
    for (int k = part; k < nc; k += HT) {
        const float pk = expf(z[k] - mx) * inv, yk = y[k];
        ce -= yk * logf(a.eps_ce / (float)nc + (1.f - a.eps_ce) * pk);
        if (pk > pmax) { pmax = pk; ap = k; }                 // (ascending k: the first maximum of this thread's classes)
        if (yk > ymax) { ymax = yk; ay = k; }
    }
    ce = quad_sum(ce);
#pragma unroll
    for (int m = 1; m < HT; m <<= 1) {                         // arg-max over the sample's threads: larger value, then smaller index (tf.argmax)
        const float p2 = __shfl_xor(pmax, m), y2 = __shfl_xor(ymax, m);
        const int ap2 = __shfl_xor(ap, m), ay2 = __shfl_xor(ay, m);
        if (p2 > pmax || (p2 == pmax && ap2 < ap)) { pmax = p2; ap = ap2; }
        if (y2 > ymax || (y2 == ymax && ay2 < ay)) { ymax = y2; ay = ay2; }
    }
    if (ok && part == 0) { a.c_err[r] = ce; a.d_cor[r] = ap == ay ? 1.f : 0.f; }
}

// grid: wpe workgroups per exit -- workgroup 0 the router tail (the whole batch: its BatchNorms need every sample),
// workgroups 1 .. the head, TT / HT samples each
__global__ __launch_bounds__(TT) void exit_tail_fwd_gen_k(const mpnn_exit_tail_args *__restrict__ tab, const int wpe) {
    const int rec = blockIdx.x / wpe, role = blockIdx.x - rec * wpe;
    const mpnn_exit_tail_args &a = tab[rec];
    const int n = a.n, tid = threadIdx.x;
    if (role > 0) {
        if (rec == 0 && role == 1) {            // (the accumulators mpnn_route adds to, this step's schedule values: see mpnn_exit_tail_args)
            if (a.clear_f) for (int i = tid; i < a.n_clear_f; i += TT) a.clear_f[i] = 0.f;
            if (a.clear_d) for (int i = tid; i < a.n_clear_d; i += TT) a.clear_d[i] = 0.0;
            if (a.hyp_src && tid < MPNN_HYP_N) a.hyp_dst[tid] = a.hyp_src[tid];
        }
        if (a.z && (role - 1) * (TT / HT) < n) gen_head_fwd(a, (role - 1) * (TT / HT));
        return;
    }
    if (!a.h1) return;
    __shared__ RouterStat st;
    __shared__ double lds[2 * TT];
    __shared__ float a1L[16 * GEN_R];
    const int R = a.R, R2 = a.R2 > 0 ? a.R2 : a.R, S = a.n_sinks;
    const bool batch = a.mode == MPNN_ACT_BN_BATCH;
    const float inv_n = 1.f / (float)n, d = a.bn_decay, d2 = a.bn_decay2;
    // ---- first BatchNorm: sum and sum of squares in fp64 (one pass: exact enough for fp32 inputs), biased variance,
    //      moving averages ----
    if (batch) {
        col_sums<2>(n, R, lds, [&](int r, int c, float *v) { const float x = a.h1[(size_t)r * R + c]; v[0] = x; v[1] = x * x; },
                    [&](int c, const double *t) {
                        const double mu = t[0] / n, vd = t[1] / n - mu * mu;
                        const float var = (float)(vd > 0.0 ? vd : 0.0);
                        st.m1[c] = (float)mu; st.s1[c] = rsqrtf(var + a.bn_eps);
                        a.m1[c] = d * a.m1[c] + (1.f - d) * (float)mu;
                        a.v1[c] = d * a.v1[c] + (1.f - d) * var;
                    });
    } else {
        for (int c = tid; c < R; c += TT) { st.m1[c] = a.m1[c]; st.s1[c] = rsqrtf(a.v1[c] + a.bn_eps); }
        __syncthreads();
    }
    // ---- h2 = relu(bn1(h1)) @ w2 + bias2, RC rows at a time through LDS (the whole batch at once for R = 32, n = 128) ----
    const int RC = (16 * GEN_R) / R;
    for (int r0 = 0; r0 < n; r0 += RC) {
        const int rows = n - r0 < RC ? n - r0 : RC;
        for (int e = tid; e < rows * R; e += TT) {
            const int rr = e / R, c = e - rr * R;
            a1L[e] = fmaxf(a.g1[c] * (a.h1[(size_t)r0 * R + e] - st.m1[c]) * st.s1[c] + a.b1[c], 0.f);
        }
        __syncthreads();
        for (int e = tid; e < rows * R2; e += TT) {
            const int rr = e / R2, j = e - rr * R2;
            float h = a.bias2[j];
            for (int c = 0; c < R; ++c) h += a1L[rr * R + c] * a.w2[c * R2 + j];
            a.h2[(size_t)r0 * R2 + e] = h;
        }
        __syncthreads();
    }
    phase_barrier();
    // ---- second BatchNorm ----
    if (batch) {
        col_sums<2>(n, R2, lds, [&](int r, int j, float *v) { const float x = ld_l2(a.h2 + (size_t)r * R2 + j); v[0] = x; v[1] = x * x; },
                    [&](int j, const double *t) {
                        const double mu = t[0] / n, vd = t[1] / n - mu * mu;
                        const float var = (float)(vd > 0.0 ? vd : 0.0);
                        st.m2[j] = (float)mu; st.s2[j] = rsqrtf(var + a.bn_eps2);
                        a.m2[j] = d2 * a.m2[j] + (1.f - d2) * (float)mu;
                        a.v2[j] = d2 * a.v2[j] + (1.f - d2) * var;
                    });
    } else {
        for (int j = tid; j < R2; j += TT) { st.m2[j] = a.m2[j]; st.s2[j] = rsqrtf(a.v2[j] + a.bn_eps2); }
        __syncthreads();
    }
    (void)inv_n;
    if (a.bn_save)
        for (int c = tid; c < 2 * R + 2 * R2; c += TT)
            a.bn_save[c] = c < R ? st.m1[c] : c < 2 * R ? st.s1[c - R] : c < 2 * R + R2 ? st.m2[c - 2 * R] : st.s2[c - 2 * R - R2];
    // ---- r = relu(bn2(h2)) @ w3 + bias3: a thread per (sample, sink) ----
    for (int e = tid; e < n * S; e += TT) {
        const int r = e / S, s = e - r * S;
        float out = a.bias3[s];
        for (int j = 0; j < R2; ++j) {
            const float a2 = fmaxf(a.g2[j] * (ld_l2(a.h2 + (size_t)r * R2 + j) - st.m2[j]) * st.s2[j] + a.b2[j], 0.f);
            out += a2 * a.w3[j * S + s];
        }
        a.r[(size_t)r * a.r_stride + s] = out;
    }
}

// ---------------------------------------------------------------------------
// Backward of the same (training mode): the derivation is in exit_tail.hip.  With xh = (h - m) rstd of a BatchNorm and
// dy the gradient w.r.t. its output behind the ReLU mask: dh = g rstd (dy - mean(dy) - xh mean(dy xh)).
//   dy2[r][j] = [a2 > 0] sum_s dr[r][s] w3[j][s]      (recomputed where needed: S <= 4 terms)
//   dh2 -> the scratch map b.dh2 [n, R2];  dW2 = a1' dh2;  dy1 = [a1 > 0] dh2 w2' -> the dh1 output, then dh1 in place
// ---------------------------------------------------------------------------
__device__ __forceinline__ void gen_head_bwd(const mpnn_exit_tail_bwd_args &b, int r0) {
    const mpnn_exit_tail_args &a = b.f;
    const int tid = threadIdx.x, r = r0 + tid / HT, part = tid % HT, nc = a.n_cls;
    const bool ok = r < a.n;
    const float *z = a.z + (size_t)(ok ? r : 0) * nc, *y = a.y + (size_t)(ok ? r : 0) * nc;
    float mx = -3.0e38f;
    for (int k = part; k < nc; k += HT) mx = fmaxf(mx, z[k]);
    mx = quad_max(mx);
    float sum = 0.f;
    for (int k = part; k < nc; k += HT) sum += expf(z[k] - mx);
    sum = quad_sum(sum);
    const float inv = 1.f / sum, hw = b.w_cerr[ok ? r : 0];
    float dot = 0.f;
    for (int k = part; k < nc; k += HT) {
        const float p = expf(z[k] - mx) * inv, q = a.eps_ce / (float)nc + (1.f - a.eps_ce) * p;
        dot += -hw * y[k] * (1.f - a.eps_ce) / q * p;
    }
    dot = quad_sum(dot);
    if (!ok) return;
    for (int k = part; k < nc; k += HT) {
        const float p = expf(z[k] - mx) * inv, q = a.eps_ce / (float)nc + (1.f - a.eps_ce) * p;
        b.dz[(size_t)r * nc + k] = p * (-hw * y[k] * (1.f - a.eps_ce) / q - dot);
    }
}

__global__ __launch_bounds__(TT) void exit_tail_bwd_gen_k(const mpnn_exit_tail_bwd_args *__restrict__ tab, const int wpe) {
    const int rec = blockIdx.x / wpe, role = blockIdx.x - rec * wpe;
    const mpnn_exit_tail_bwd_args &b = tab[rec];
    const mpnn_exit_tail_args &a = b.f;
    const int n = a.n, tid = threadIdx.x;
    if (role > 0) {
        if (a.z && b.dz && (role - 1) * (TT / HT) < n) gen_head_bwd(b, (role - 1) * (TT / HT));
        return;
    }
    if (!a.h1) return;
    const int R = a.R, R2 = a.R2 > 0 ? a.R2 : a.R, S = a.n_sinks;
    __shared__ RouterStat st;
    // one arena: the column-sum exchange (6 x TT doubles) and the row-chunk tiles of the dW2 pass are never live together
    __shared__ double lds[6 * TT];
    float *a1L = (float *)lds, *dhL = a1L + 16 * GEN_R;
    for (int c = tid; c < 2 * R + 2 * R2; c += TT) {
        const float v = a.bn_save[c];
        if (c < R) st.m1[c] = v; else if (c < 2 * R) st.s1[c - R] = v; else if (c < 2 * R + R2) st.m2[c - 2 * R] = v; else st.s2[c - 2 * R - R2] = v;
    }
    if (tid >= TT - 64) {                         // the last wave, beside the table loads: dbias3[s] = sum_r dr[r][s]
        const int lane = tid & 63;
        for (int s = 0; s < S; ++s) {
            float acc = 0.f;
            for (int r = lane; r < n; r += 64) acc += b.dr[(size_t)r * a.r_stride + s];
            acc = wave_sum_f(acc);
            if (lane == 0) b.dbias3[s] = acc;
        }
    }
    __syncthreads();
    const float inv = 1.f / (float)n;
    // dy2 and xhat2 of one element
    auto dy2 = [&](int r, int j, float &xh) {
        const float h = a.h2[(size_t)r * R2 + j];
        xh = (h - st.m2[j]) * st.s2[j];
        float dsum = 0.f;
        for (int s = 0; s < S; ++s) dsum += b.dr[(size_t)r * a.r_stride + s] * a.w3[j * S + s];
        return (a.g2[j] * xh + a.b2[j] <= 0.f) ? 0.f : dsum;
    };
    // column j of the second BatchNorm: dbeta2, dgamma2, the two means, the dW3 row
    col_sums<6>(n, R2, lds,
                [&](int r, int j, float *v) {
                    float xh;
                    const float dy = dy2(r, j, xh);
                    v[0] = dy; v[1] = dy * xh;
                    const float a2 = fmaxf(a.g2[j] * xh + a.b2[j], 0.f);
#pragma unroll
                    for (int s = 0; s < MPNN_MAX_SINKS; ++s) v[2 + s] = s < S ? a2 * b.dr[(size_t)r * a.r_stride + s] : 0.f;
                },
                [&](int j, const double *t) {
                    b.db2[j] = (float)t[0]; b.dg2[j] = (float)t[1];
                    st.p2[j] = (float)t[0] * inv; st.q2[j] = (float)t[1] * inv;
                    for (int s = 0; s < S; ++s) b.dw3[j * S + s] = (float)t[2 + s];
                });
    // dh2 -> scratch, and its column sums (dbias2) in the same pass
    col_sums<1>(n, R2, lds,
                [&](int r, int j, float *v) {
                    float xh;
                    const float dy = dy2(r, j, xh);
                    const float dh = a.g2[j] * st.s2[j] * (dy - st.p2[j] - xh * st.q2[j]);
                    b.dh2[(size_t)r * R2 + j] = dh;
                    v[0] = dh;
                },
                [&](int j, const double *t) { b.dbias2[j] = (float)t[0]; });
    phase_barrier();
    // dW2[c][j] = sum_r a1[r][c] dh2[r][j]: (c, j) pairs, 2 per thread and pass, row chunks of 16 through LDS
    const int npair = R * R2, RC = (16 * GEN_R) / (R > R2 ? R : R2);
    for (int pb = 0; pb < npair; pb += 2 * TT) {
        float acc[2];
        int pc[2], pj[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int p = pb + tid + q * TT, pq = p < npair ? p : 0;
            acc[q] = 0.f; pc[q] = pq / R2; pj[q] = pq - pc[q] * R2;
        }
        for (int r0 = 0; r0 < n; r0 += RC) {
            const int rows = n - r0 < RC ? n - r0 : RC;
            __syncthreads();
            for (int e = tid; e < rows * R; e += TT) {
                const int rr = e / R, c = e - rr * R;
                a1L[e] = fmaxf(a.g1[c] * (a.h1[(size_t)r0 * R + e] - st.m1[c]) * st.s1[c] + a.b1[c], 0.f);
            }
            for (int e = tid; e < rows * R2; e += TT) dhL[e] = ld_l2(b.dh2 + (size_t)r0 * R2 + e);
            __syncthreads();
            for (int rr = 0; rr < rows; ++rr)
#pragma unroll
                for (int q = 0; q < 2; ++q) acc[q] += a1L[rr * R + pc[q]] * dhL[rr * R2 + pj[q]];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) { const int p = pb + tid + q * TT; if (p < npair) b.dw2[p] = acc[q]; }
    }
    __syncthreads();
    // dy1[r][c] = [a1 > 0] sum_j dh2[r][j] w2[c][j] -> the dh1 output (turned into dh1 below), and the first BatchNorm's
    // column sums in the same pass
    // (w2 through LDS at pitch R2 + 1 when it fits beside the exchange area: a thread walks ROW c of w2, so across the
    // lanes the global loads would touch a cache line each)
    float *w2L = (float *)(lds + 2 * TT);
    const bool w2_lds = R * (R2 + 1) <= 8192;
    if (w2_lds) for (int e = tid; e < R * R2; e += TT) { const int c = e / R2, j = e - c * R2; w2L[c * (R2 + 1) + j] = a.w2[e]; }
    __syncthreads();
    const float *w2p = w2_lds ? w2L : a.w2;
    const int w2s = w2_lds ? R2 + 1 : R2;
    col_sums<2>(n, R, lds,
                [&](int r, int c, float *v) {
                    const float xh = (a.h1[(size_t)r * R + c] - st.m1[c]) * st.s1[c];
                    float dsum = 0.f;
                    if (a.g1[c] * xh + a.b1[c] > 0.f)
                        for (int j = 0; j < R2; ++j) dsum += ld_l2(b.dh2 + (size_t)r * R2 + j) * w2p[c * w2s + j];
                    b.dh1[(size_t)r * R + c] = dsum;
                    v[0] = dsum; v[1] = dsum * xh;
                },
                [&](int c, const double *t) {
                    b.db1[c] = (float)t[0]; b.dg1[c] = (float)t[1];
                    st.p1[c] = (float)t[0] * inv; st.q1[c] = (float)t[1] * inv;
                });
    phase_barrier();
    for (int e = tid; e < n * R; e += TT) {
        const int r = e / R, c = e - r * R;
        const float xh = (a.h1[e] - st.m1[c]) * st.s1[c];
        b.dh1[e] = a.g1[c] * st.s1[c] * (ld_l2(b.dh1 + e) - st.p1[c] - xh * st.q1[c]);
    }
}

// ---------------------------------------------------------------------------
// evaluation (mpnn_exit_ev_gen; moving-average BatchNorms: every sample on its own).  Two launches:
//   ev_lin_gen_k   the two affine maps over the exit's features on MFMA tiles (gen_lin_tile), on the record's sample
//                  list (idx / device-side cnt), into the record's scratch maps z [n, n_cls] and h1 [n, R] (by image);
//   ev_tail_gen_k  64 samples per workgroup, HT threads per sample: the router tail through LDS rows, the head's softmax /
//                  cross-entropy / arg-max, and the children's sample lists -- ONE reservation (atomic) per workgroup
//                  and sink, then every sample writes itself at base + its rank within the workgroup.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ev_lin_gen_k(const mpnn_exit_ev_args *__restrict__ tab) {
    const mpnn_exit_ev_args &a = tab[blockIdx.z];
    int n = a.n;
    if (a.cnt) { const int c = *a.cnt; n = c < n ? c : n; }
    const int n0 = blockIdx.x * 16;
    if (n0 >= n) return;
    const int M0 = a.w_head ? a.n_cls : 0, M1 = a.w1 ? a.R : 0;
    const int T0 = (M0 + 15) >> 4, T1 = (M1 + 15) >> 4;
    int ct = blockIdx.y;
    if (ct >= T0 + T1) return;
    const int s = ct >= T0 ? 1 : 0;
    if (s) ct -= T0;
    __shared__ float cA[GEN_C * 3];
    __shared__ float red[4 * 256];
    if ((s && !a.h1) || (!s && !a.z)) return;       // (a record without its scratch map: nothing to write to)
    if (s) gen_lin_tile(a.a, a.HW, n, a.idx, n0, a.w1, a.b1, M1, ct * 16, a.extra_col != 0, a.alpha_cpt, a.k_cpt, a.h1, cA, red);
    else   gen_lin_tile(a.a, a.HW, n, a.idx, n0, a.w_head, a.b_head, M0, ct * 16, false, 0.f, nullptr, a.z, cA, red);
}

#define EV_SPW 64            // samples per workgroup of the tail
__global__ __launch_bounds__(256) void ev_tail_gen_k(const mpnn_exit_ev_args *__restrict__ tab) {
    const mpnn_exit_ev_args &a = tab[blockIdx.y];
    int n = a.n;
    if (a.cnt) { const int c = *a.cnt; n = c < n ? c : n; }
    const int s0 = blockIdx.x * EV_SPW;
    if (s0 >= n) return;
    __shared__ float aL[16 * GEN_R], bL[16 * GEN_R];
    __shared__ int arg_s[EV_SPW], img_s[EV_SPW], base_s[MPNN_MAX_SINKS];
    const int tid = threadIdx.x, sl = tid / HT, part = tid % HT;
    const int nc = (a.w_head && a.z) ? a.n_cls : 0, R = (a.w1 && a.h1) ? a.R : 0, R2 = R ? (a.R2 > 0 ? a.R2 : a.R) : 0, S = R ? a.n_sinks : 0;
    if (tid < EV_SPW) { arg_s[tid] = -1; img_s[tid] = -1; }
    __syncthreads();
    for (int it = 0; it < EV_SPW / 16; ++it) {
        const int slot = s0 + it * 16 + sl;
        const bool ok = slot < n;
        const int img = ok ? (a.idx ? a.idx[slot] : slot) : 0;
        if (R) {
            for (int c = part; c < R; c += HT)
                aL[sl * R + c] = fmaxf(a.g1[c] * (a.h1[(size_t)img * R + c] - a.m1[c]) * rsqrtf(a.v1[c] + a.bn_eps) + a.be1[c], 0.f);
            __syncthreads();
            for (int j = part; j < R2; j += HT) {
                float h = a.bias2[j];
                for (int c = 0; c < R; ++c) h += aL[sl * R + c] * a.w2[c * R2 + j];
                bL[sl * R2 + j] = fmaxf(a.g2[j] * (h - a.m2[j]) * rsqrtf(a.v2[j] + a.bn_eps2) + a.be2[j], 0.f);
            }
            __syncthreads();
            if (part == 0 && ok) {
                int arg = 0; float rmax = 0.f;
                for (int s = 0; s < S; ++s) {
                    float r = a.bias3[s];
                    for (int j = 0; j < R2; ++j) r += bL[sl * R2 + j] * a.w3[j * S + s];
                    a.r[(size_t)img * a.r_stride + s] = r;
                    if (s == 0 || r > rmax) { rmax = r; arg = s; }             // first index on ties (tf.argmax)
                }
                arg_s[it * 16 + sl] = arg; img_s[it * 16 + sl] = img;
            }
        }
        if (nc) {            // the head: HT threads per sample (as gen_head_fwd, by image)
            const float *z = a.z + (size_t)img * nc, *y = a.y + (size_t)img * nc;
            float mx = -3.0e38f;
            for (int k = part; k < nc; k += HT) mx = fmaxf(mx, z[k]);
            mx = quad_max(mx);
            float sum = 0.f;
            for (int k = part; k < nc; k += HT) sum += expf(z[k] - mx);
            sum = quad_sum(sum);
            const float inv = 1.f / sum;
            float ce = 0.f, pmax = -1.f, ymax = -3.0e38f; int ap = nc, ay = nc;
            for (int k = part; k < nc; k += HT) {
                const float pk = expf(z[k] - mx) * inv, yk = y[k];
                ce -= yk * logf(a.eps_ce / (float)nc + (1.f - a.eps_ce) * pk);
                if (pk > pmax) { pmax = pk; ap = k; }
                if (yk > ymax) { ymax = yk; ay = k; }
            }
            ce = quad_sum(ce);
#pragma unroll
            for (int m = 1; m < HT; m <<= 1) {
                const float p2 = __shfl_xor(pmax, m), y2 = __shfl_xor(ymax, m);
                const int ap2 = __shfl_xor(ap, m), ay2 = __shfl_xor(ay, m);
                if (p2 > pmax || (p2 == pmax && ap2 < ap)) { pmax = p2; ap = ap2; }
                if (y2 > ymax || (y2 == ymax && ay2 < ay)) { ymax = y2; ay = ay2; }
            }
            if (ok && part == 0) { a.c_err[img] = ce; a.d_cor[img] = ap == ay ? 1.f : 0.f; }
        }
        __syncthreads();
    }
    if (!S) return;
    // the children's lists: one reservation per (workgroup, sink), ranks within the workgroup in slot order
    if (tid < MPNN_MAX_SINKS) {
        int cnt = 0;
        for (int k = 0; k < EV_SPW; ++k) cnt += arg_s[k] == tid ? 1 : 0;
        base_s[tid] = (cnt > 0 && tid < S && a.child_idx[tid]) ? atomicAdd(a.child_cnt[tid], cnt) : -1;
    }
    __syncthreads();
    if (tid < EV_SPW && arg_s[tid] >= 0) {
        const int arg = arg_s[tid], base = base_s[arg];
        if (base >= 0) {
            int rank = 0;
            for (int k = 0; k < tid; ++k) rank += arg_s[k] == arg ? 1 : 0;
            if (base + rank < a.n) a.child_idx[arg][base + rank] = img_s[tid];
        }
    }
}

// Widest head / router the records of the process need (column tiles of the forward map's grid): the records live in
// device memory, so the launcher cannot read them -- mpnn_exit_gen_check, which the caller runs on every exit before
// uploading its record, keeps the maxima.  (Workgroups beyond a record's own tiles return at once.)
// (a process-wide MAXIMUM that only grows, kept with an atomic max: engines that validate their exits from different host
// threads cannot lose each other's larger value; a launch with more column tiles than a record needs costs workgroups that
// return at once, never correctness)
static std::atomic<int> g_gen_tiles{2};

extern "C" int mpnn_lin_fwd_gen(const mpnn_lin_fwd_args *dev_table, int count, int n_max, void *stream) {
    if (count <= 0 || n_max <= 0) return 0;
    if (!dev_table) return MPNN_E_ARG;
    hipLaunchKernelGGL(lin_fwd_gen_k, dim3((n_max + 15) / 16, g_gen_tiles.load(), count), dim3(256), 0, (hipStream_t)stream, dev_table);
    MPNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mpnn_lin_bwd_gen(const mpnn_lin_bwd_args *dev_table, int count, int n_max, int k_max, void *stream) {
    if (count <= 0 || n_max <= 0) return 0;
    if (!dev_table || k_max <= 0) return MPNN_E_ARG;
    hipLaunchKernelGGL(lin_bwd_gen_k, dim3((k_max + 2 + 15) / 16, count, 2), dim3(256), 0, (hipStream_t)stream, dev_table);
    MPNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mpnn_exit_tail_fwd_gen(const mpnn_exit_tail_args *dev_table, int count, int n_max, void *stream) {
    if (count <= 0 || n_max <= 0) return 0;
    if (!dev_table) return MPNN_E_ARG;
    const int wpe = 1 + (n_max + TT / HT - 1) / (TT / HT);
    hipLaunchKernelGGL(exit_tail_fwd_gen_k, dim3(count * wpe), dim3(TT), 0, (hipStream_t)stream, dev_table, wpe);
    MPNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mpnn_exit_tail_bwd_gen(const mpnn_exit_tail_bwd_args *dev_table, int count, int n_max, void *stream) {
    if (count <= 0 || n_max <= 0) return 0;
    if (!dev_table) return MPNN_E_ARG;
    const int wpe = 1 + (n_max + TT / HT - 1) / (TT / HT);
    hipLaunchKernelGGL(exit_tail_bwd_gen_k, dim3(count * wpe), dim3(TT), 0, (hipStream_t)stream, dev_table, wpe);
    MPNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mpnn_exit_ev_gen(const mpnn_exit_ev_args *dev_table, int count, int n_max, void *stream) {
    if (count <= 0 || n_max <= 0) return 0;
    if (!dev_table) return MPNN_E_ARG;
    hipLaunchKernelGGL(ev_lin_gen_k, dim3((n_max + 15) / 16, g_gen_tiles.load(), count), dim3(256), 0, (hipStream_t)stream, dev_table);
    MPNN_LAUNCH_CHECK();
    hipLaunchKernelGGL(ev_tail_gen_k, dim3((n_max + EV_SPW - 1) / EV_SPW, count), dim3(256), 0, (hipStream_t)stream, dev_table);
    MPNN_LAUNCH_CHECK();
    return 0;
}

// host-side limits of the any-width forms (records live in device memory: the caller validates before uploading)
extern "C" int mpnn_exit_gen_check(int C, int K, int n_cls, int R, int R2, int n_sinks) {
    if (C < 1 || C > GEN_C || K < 1 || K > GEN_K || (K % C)) return MPNN_E_SHAPE;
    if (n_cls < 0 || n_cls > 1024 || R < 0 || R > GEN_R || R2 < 0 || R2 > GEN_R) return MPNN_E_SHAPE;
    if (R && (n_sinks < 2 || n_sinks > MPNN_MAX_SINKS)) return MPNN_E_SHAPE;
    const int tiles = (n_cls + 15) / 16 + (R + 15) / 16;
    for (int cur = g_gen_tiles.load(); tiles > cur && !g_gen_tiles.compare_exchange_weak(cur, tiles);) {}
    return 0;
}
