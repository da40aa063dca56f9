// Exit-path affine maps: the LogReg head's LinTrans and the router's first
// LinTrans share one pass over the block's coarsest scale (BatchNorm + ReLU
// applied on load).  Forward on v_mfma_f32_16x16x4_f32 (16 samples x 16
// outputs per tile, K split over the 4 waves); backward as two small
// HBM/L2-bound kernels (dX and dW), deterministic (no atomics).
#include "common.h"

// ------------------------------- forward ------------------------------------
__global__ __launch_bounds__(256) void lin_fwd_k(const mpnn_lin_fwd_args *__restrict__ tab) {
    const mpnn_lin_fwd_args &a = tab[blockIdx.y];
    const int n0 = blockIdx.x * 16;
    if (n0 >= a.n) return;
    __shared__ float cA[128 * 3];
    __shared__ float red[4 * 2 * 256];
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63, g = lane >> 4, li = lane & 15;
    const int C = a.a.C, K = a.HW * C;
    if (a.a.mode != MPNN_ACT_IDENTITY) {
        for (int c = tid; c < C; c += 256) {
            const BnC k = bn_coef(a.a, c);
            cA[c * 3] = k.m; cA[c * 3 + 1] = k.gamma * k.rstd; cA[c * 3 + 2] = k.beta;
        }
    }
    __syncthreads();
    const int row = n0 + li;
    const bool valid = row < a.n;
    f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    for (int kb = wid; kb < (K >> 4); kb += 4) {
        const int k = kb * 16 + 4 * g;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (valid) {
            x = *(const f32x4 *)(a.a.x + (size_t)row * K + k);
            if (a.a.mode != MPNN_ACT_IDENTITY) {
                const int c = k % C;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float *cc = cA + (c + j) * 3;
                    x[j] = fmaxf((x[j] - cc[0]) * cc[1] + cc[2], 0.f);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (!a.w[s]) continue;
            const int M = a.M[s];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float b = li < M ? a.w[s][(size_t)(k + j) * M + li] : 0.f;
                acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[j], b, acc[s], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wid * 2 + s) * 256 + lane * 4 + r] = acc[s][r];
    __syncthreads();
    if (tid < 64) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (!a.w[s]) continue;
            const int M = a.M[s];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int orow = n0 + (tid >> 4) * 4 + r, col = tid & 15;
                if (orow < a.n && col < M) {
                    float v = a.b[s][col];
#pragma unroll
                    for (int w = 0; w < 4; ++w) v += red[(w * 2 + s) * 256 + tid * 4 + r];
                    if (a.extra_col[s]) v += a.alpha_cpt * a.k_cpt[orow] * a.w[s][(size_t)K * M + col];
                    a.y[s][(size_t)orow * M + col] = v;
                }
            }
        }
    }
}

extern "C" int mpnn_lin_fwd(const mpnn_lin_fwd_args *dev_table, int count, int n_max, void *stream) {
    if (count <= 0 || n_max <= 0) return 0;
    if (!dev_table) return MPNN_E_ARG;
    hipLaunchKernelGGL(lin_fwd_k, dim3((n_max + 15) / 16, count), dim3(256), 0, (hipStream_t)stream, dev_table);
    MPNN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------- backward -----------------------------------
// dX[row][k] = sum_s sum_m dy_s[row][m] * w_s[k][m]        (one thread per 4 k)
__global__ __launch_bounds__(256) void lin_bwd_dx_k(const mpnn_lin_bwd_args *__restrict__ tab) {
    const mpnn_lin_bwd_args &a = tab[blockIdx.z];
    if (!a.dx) return;
    const int K = a.HW * a.a.C;
    const int row = blockIdx.y;
    const int k = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (row >= a.n || k >= K) return;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (!a.w[s]) continue;
        const int M = a.M[s];
        const float *dy = a.dy[s] + (size_t)row * M;
        for (int m = 0; m < M; ++m) {
            const float d = dy[m];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += d * a.w[s][(size_t)(k + j) * M + m];
        }
    }
    *(f32x4 *)(a.dx + (size_t)row * K + k) = acc;
}

// dW_s[k][m] = sum_row act(X)[row][k] * dy_s[row][m]; db_s[m] = sum_row dy_s[row][m]
__global__ __launch_bounds__(256) void lin_bwd_dw_k(const mpnn_lin_bwd_args *__restrict__ tab) {
    const mpnn_lin_bwd_args &a = tab[blockIdx.y];
    const int C = a.a.C, K = a.HW * C;
    const int kext = K + ((a.extra_col[0] || a.extra_col[1]) ? 1 : 0);
    if ((int)(blockIdx.x * 256) >= kext) return;
    __shared__ float dys[64 * 32];
    const int k = blockIdx.x * 256 + threadIdx.x;
    float cm = 0.f, ca = 1.f, cb = 0.f;
    const bool bn = a.a.mode != MPNN_ACT_IDENTITY;
    if (bn && k < K) { const BnC c = bn_coef(a.a, k % C); cm = c.m; ca = c.gamma * c.rstd; cb = c.beta; }
    float acc[2][16];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int m = 0; m < 16; ++m) acc[s][m] = 0.f;
    float dbs = 0.f;
    for (int r0 = 0; r0 < a.n; r0 += 64) {
        const int nr = min(64, a.n - r0);
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * 32; i += 256) {
            const int rr = i >> 5, col = i & 31, s = col >> 4, m = col & 15;
            dys[i] = (rr < nr && a.w[s] && m < a.M[s]) ? a.dy[s][(size_t)(r0 + rr) * a.M[s] + m] : 0.f;
        }
        __syncthreads();
        if (blockIdx.x == 0 && threadIdx.x < 32)
            for (int rr = 0; rr < nr; ++rr) dbs += dys[rr * 32 + threadIdx.x];
        if (k < kext) {
            for (int rr = 0; rr < nr; ++rr) {
                float x;
                if (k < K) {
                    x = a.a.x[(size_t)(r0 + rr) * K + k];
                    if (bn) x = fmaxf((x - cm) * ca + cb, 0.f);
                } else {
                    x = a.alpha_cpt * a.k_cpt[r0 + rr];
                }
#pragma unroll
                for (int col = 0; col < 32; ++col) acc[col >> 4][col & 15] += x * dys[rr * 32 + col];
            }
        }
    }
    if (k < kext) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (!a.w[s] || !a.dw[s]) continue;
            if (k == K && !a.extra_col[s]) continue;
            const int M = a.M[s];
#pragma unroll
            for (int m = 0; m < 16; ++m) if (m < M) a.dw[s][(size_t)k * M + m] = acc[s][m];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < 32) {
        const int s = threadIdx.x >> 4, m = threadIdx.x & 15;
        if (a.w[s] && a.db[s] && m < a.M[s]) a.db[s][m] = dbs;
    }
}

extern "C" int mpnn_lin_bwd(const mpnn_lin_bwd_args *dev_table, int count, int n_max, int k_max, void *stream) {
    if (count <= 0 || n_max <= 0) return 0;
    if (!dev_table) return MPNN_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(lin_bwd_dx_k, dim3((k_max / 4 + 255) / 256, n_max, count), dim3(256), 0, st, dev_table);
    MPNN_LAUNCH_CHECK();
    hipLaunchKernelGGL(lin_bwd_dw_k, dim3((k_max + 1 + 255) / 256, count), dim3(256), 0, st, dev_table);
    MPNN_LAUNCH_CHECK();
    return 0;
}
