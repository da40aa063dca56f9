// Exit-path affine maps: the LogReg head's LinTrans and the router's first
// LinTrans share one pass over the block's coarsest scale (BatchNorm + ReLU
// applied on load).  These launches sit on the step's critical path (forward
// trunk -> exits -> router -> exits backward -> trunk backward) with almost no
// arithmetic, so they are organised for latency: table-driven (one launch for
// every exit of the tree), K split over 16 waves in the forward, and ONE
// backward kernel in which a thread owns a feature k, streams the batch rows
// (dY broadcast from LDS, X prefetched 16 rows at a time) and produces dX, dW
// and db in a single pass (row groups add their dW/db partials with fp32 atomics).
#include "common.h"

// ------------------------------- forward ------------------------------------
// 16 samples x 16 outputs per MFMA tile; waves split K, partial tiles meet in LDS.
#define LF_WAVES 16
__global__ __launch_bounds__(LF_WAVES * 64) void lin_fwd_k(const mpnn_lin_fwd_args *__restrict__ tab) {
    const mpnn_lin_fwd_args &a = tab[blockIdx.y];
    const int n0 = blockIdx.x * 16;
    if (n0 >= a.n) return;
    trace_stamp(0); trace_note(6, 10);
    __shared__ float cA[128 * 3];
    __shared__ float red[LF_WAVES * 2 * 256];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = a.a.C, K = a.HW * C;
    const bool bn = a.a.mode != MPNN_ACT_IDENTITY;
    if (bn) {
        for (int c = tid; c < C; c += LF_WAVES * 64) {
            const BnC k = bn_coef(a.a, c);
            cA[c * 3] = k.m; cA[c * 3 + 1] = k.gamma * k.rstd; cA[c * 3 + 2] = k.beta;
        }
    }
    __syncthreads();
    trace_stamp(1);
    const int row = n0 + li;
    const bool valid = row < a.n;
    const int M0 = a.w[0] ? a.M[0] : 0, M1 = a.w[1] ? a.M[1] : 0;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    // LF_UN 16-feature blocks per iteration, ALL their loads issued before the first MFMA: a wave's K
    // share is a chain of dependent memory round trips (one per iteration), 8 of them at K = 2048 with
    // one block per iteration -- two with four.
    constexpr int LF_UN = 4;
    const int nkb = K >> 4;
    const float *xrow = a.a.x + (size_t)(valid ? row : 0) * K;
    for (int kb = wid * LF_UN; kb < nkb; kb += LF_WAVES * LF_UN) {
        f32x4 x[LF_UN];
        float b0[LF_UN][4], b1[LF_UN][4];
#pragma unroll
        for (int u = 0; u < LF_UN; ++u) {
            const int kk = kb + u < nkb ? kb + u : kb;          // (past the end: a repeat of the first block, masked below)
            const int k = kk * 16 + 4 * g;
            x[u] = *(const f32x4 *)(xrow + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                b0[u][j] = li < M0 ? a.w[0][(size_t)(k + j) * M0 + li] : 0.f;
                b1[u][j] = li < M1 ? a.w[1][(size_t)(k + j) * M1 + li] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < LF_UN; ++u) {
            const bool on = valid && kb + u < nkb;
            const int k = (kb + u) * 16 + 4 * g;
            if (bn) {
                const int c = k % C;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float *cc = cA + (c + j) * 3;
                    x[u][j] = fmaxf((x[u][j] - cc[0]) * cc[1] + cc[2], 0.f);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xv = on ? x[u][j] : 0.f;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv, b0[u][j], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv, b1[u][j], acc1, 0, 0, 0);
            }
        }
    }
    trace_stamp(4);
    mfma_drain();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[(wid * 2 + 0) * 256 + lane * 4 + r] = acc0[r];
        red[(wid * 2 + 1) * 256 + lane * 4 + r] = acc1[r];
    }
    __syncthreads();
    if (tid < 512) {                         // (set, lane, r): 2 * 64 * 4 outputs
        const int s = tid >> 8, e = tid & 255, l = e >> 2, r = e & 3;
        if (a.w[s]) {
            const int M = a.M[s];
            const int orow = n0 + (l >> 4) * 4 + r, col = l & 15;
            if (orow < a.n && col < M) {
                float v = a.b[s][col];
#pragma unroll
                for (int w = 0; w < LF_WAVES; ++w) v += red[(w * 2 + s) * 256 + e];
                if (a.extra_col[s]) v += a.alpha_cpt * a.k_cpt[orow] * a.w[s][(size_t)K * M + col];
                a.y[s][(size_t)orow * M + col] = v;
            }
        }
    }
    trace_stamp(5);
}

int mpnn_trace_install_lin(void *buf) { return mpnn_trace_install(buf); }

extern "C" int mpnn_lin_fwd(const mpnn_lin_fwd_args *dev_table, int count, int n_max, void *stream) {
    if (count <= 0 || n_max <= 0) return 0;
    if (!dev_table) return MPNN_E_ARG;
    hipLaunchKernelGGL(lin_fwd_k, dim3((n_max + 15) / 16, count), dim3(LF_WAVES * 64), 0, (hipStream_t)stream,
                       dev_table);
    MPNN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------- backward -----------------------------------
// Thread <-> feature k.  For every batch row r:
//   dX[r][k]   = sum_s sum_m dY_s[r][m] * W_s[k][m]
//   dW_s[k][m] += act(X)[r][k] * dY_s[r][m]           db_s[m] = sum_r dY_s[r][m]
// W rows live in registers, dY rows are staged in LDS (<= LB_ROWS rows per pass).
#ifndef LB_ROWS
#define LB_ROWS 16          // rows per pass and per z-slice (the row loop is VALU-bound: ~100 instructions per row)
#endif
__global__ __launch_bounds__(256) void lin_bwd_k(const mpnn_lin_bwd_args *__restrict__ tab) {
    const mpnn_lin_bwd_args &a = tab[blockIdx.y];
    const int C = a.a.C, K = a.HW * C;
    const bool has_extra = a.extra_col[0] || a.extra_col[1];
    const int kext = K + (has_extra ? 1 : 0);
    if ((int)(blockIdx.x * 256) >= kext) return;
    trace_stamp(0); trace_note(6, 11);
    __shared__ float dys[LB_ROWS * 32];
    const int tid = threadIdx.x;
    const int k = blockIdx.x * 256 + tid;
    const int M0 = a.w[0] ? a.M[0] : 0, M1 = a.w[1] ? a.M[1] : 0;
    const bool bn = a.a.mode != MPNN_ACT_IDENTITY;
    float cm = 0.f, ca = 1.f, cb = 0.f, crs = 0.f;
    if (bn && k < K) { const BnC c = bn_coef(a.a, k % C); cm = c.m; ca = c.gamma * c.rstd; cb = c.beta; crs = c.rstd; }
    const bool fuse_bn = a.dz_out != nullptr && bn;          // uniform: fused mpnn_bn_bwd_reduce
    float r1 = 0.f, r2 = 0.f;                                // sum dz, sum dz * xhat of this thread's feature
    float w[32], acc[32];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        w[m] = (k < kext && m < M0 && (k < K || a.extra_col[0])) ? a.w[0][(size_t)k * M0 + m] : 0.f;
        w[16 + m] = (k < kext && m < M1 && (k < K || a.extra_col[1])) ? a.w[1][(size_t)k * M1 + m] : 0.f;
        acc[m] = 0.f; acc[16 + m] = 0.f;
    }
    trace_stamp(1);
    float dbs = 0.f;
    // blockIdx.z owns rows [z*LB_ROWS, ...) with stride gridDim.z*LB_ROWS: four times the workgroups,
    // a quarter of the serial row loop; dW/db are then ADDED into the (zeroed) gradient tensors.
    for (int r0 = blockIdx.z * LB_ROWS; r0 < a.n; r0 += gridDim.z * LB_ROWS) {
        const int nr = min(LB_ROWS, a.n - r0);
        __syncthreads();
        for (int i = tid; i < LB_ROWS * 32; i += 256) {
            const int rr = i >> 5, col = i & 31, s = col >> 4, m = col & 15;
            const int M = s ? M1 : M0;
            dys[i] = (rr < nr && m < M) ? a.dy[s][(size_t)(r0 + rr) * M + m] : 0.f;
        }
        __syncthreads();
        if (blockIdx.x == 0 && tid < 32)
            for (int rr = 0; rr < nr; ++rr) dbs += dys[rr * 32 + tid];
        if (k >= kext) continue;
        for (int rb = 0; rb < nr; rb += 16) {
            float xv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {             // 16 independent loads in flight
                const int rr = rb + u;
                xv[u] = 0.f;
                if (rr < nr) xv[u] = k < K ? a.a.x[(size_t)(r0 + rr) * K + k] : a.alpha_cpt * a.k_cpt[r0 + rr];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int rr = rb + u;
                if (rr >= nr) continue;            // (no `break`: the loop must stay fully unrolled)
                float x = xv[u];
                const float xc = x - cm;
                if (bn && k < K) x = fmaxf(xc * ca + cb, 0.f);
                const f32x4 *d4 = (const f32x4 *)(dys + rr * 32);
                float dx = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const f32x4 d = d4[q];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { acc[q * 4 + j] += x * d[j]; dx += d[j] * w[q * 4 + j]; }
                }
                if (a.dx && k < K) a.dx[(size_t)(r0 + rr) * K + k] = dx;
                if (fuse_bn && k < K) {
                    const float dz = x > 0.f ? dx : 0.f;
                    a.dz_out[(size_t)(r0 + rr) * K + k] = dz;
                    r1 += dz; r2 += dz * (xc * crs);
                }
            }
        }
    }
    // dW block of this workgroup = rows [k0, k0+256) of a [K(+1)][M] tensor: contiguous in memory.
    // Transpose the per-thread rows through LDS so every atomic wave-instruction adds 256 contiguous
    // bytes (one lane per row would put 64 lanes in 64 different 64-B segments: ~17x slower).
    trace_stamp(4);
    __shared__ float tr[256 * 17];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (!a.w[s] || !a.dw[s]) continue;          // uniform
        const int M = a.M[s];
        const int krows = a.extra_col[s] ? K + 1 : K;
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 16; ++m) tr[tid * 17 + m] = acc[s * 16 + m];
        __syncthreads();
        const int k0 = blockIdx.x * 256;
        const int rows = min(256, krows - k0);
        float *dst = a.dw[s] + (size_t)k0 * M;
        for (int i = tid; i < rows * M; i += 256) {
            const int kk = i / M, m = i - kk * M;
            atomicAdd(dst + i, tr[kk * 17 + m]);
        }
    }
    if (blockIdx.x == 0 && tid < 32) {
        const int s = tid >> 4, m = tid & 15;
        if (a.w[s] && a.db[s] && m < a.M[s]) atomicAdd(a.db[s] + m, dbs);
    }
    // fused BatchNorm-backward reductions: feature k = pixel * C + c; the workgroup's 256 features are
    // 256 / C pixels of all C channels (C | 256) or a 256-channel slice of one pixel
    if (fuse_bn) {
        __syncthreads();
        tr[tid] = r1; tr[256 + tid] = r2;
        __syncthreads();
        const int k0 = blockIdx.x * 256;
        const int span = C < 256 ? C : 256;                  // distinct channels in this workgroup
        if (tid < span && k0 + tid < K) {
            double a1 = 0.0, a2 = 0.0;
            for (int t = tid; t < 256 && k0 + t < K; t += span) { a1 += (double)tr[t]; a2 += (double)tr[256 + t]; }
            const int c = (k0 + tid) % C;
            double *slot = a.red_out + (size_t)((blockIdx.x + blockIdx.z * gridDim.x) % a.red_nslot) * 2 * C;
            atomicAdd(slot + c, a1);
            atomicAdd(slot + C + c, a2);
        }
    }
    trace_stamp(5);
}

extern "C" int mpnn_lin_bwd(const mpnn_lin_bwd_args *dev_table, int count, int n_max, int k_max, void *stream) {
    if (count <= 0 || n_max <= 0) return 0;
    if (!dev_table) return MPNN_E_ARG;
    int zsplit = (n_max + LB_ROWS - 1) / LB_ROWS;              // one pass of LB_ROWS rows per workgroup
    if (zsplit > 16) zsplit = 16;
    hipLaunchKernelGGL(lin_bwd_k, dim3((k_max + 1 + 255) / 256, count, zsplit), dim3(256), 0, (hipStream_t)stream,
                       dev_table);
    MPNN_LAUNCH_CHECK();
    return 0;
}
