// Exit tail: everything after the exit's first affine map.
//   head   : Softmax + CrossEntropyError           (layer_types.py:81-84, 262-272)
//   router : BN -> ReLU -> LinTrans(R) -> BN -> ReLU -> LinTrans(n_sinks)
//            (arch_and_hypers.py:47-49; BatchNorm over the batch, layer_types.py:219-239)
// One 256-thread workgroup per exit owns the whole batch: the router's
// BatchNorms need statistics over every sample, and at R = 16 the arithmetic
// is a few hundred KFLOP.  All reductions are two-pass (mean, then centred
// second moment) through LDS, in a fixed order: results are deterministic.
#include "common.h"

#define TR 16          // max router width
#define TS 8           // max sinks
#define TC 32          // max classes
#define CHUNK 128      // samples staged in LDS per pass

// Sum over samples of f(s)[c] for c < R; every thread returns the totals in out[c] (LDS).
// scratch: [256] floats.  Thread t owns channel t % R and samples t / R + k * (256 / R).
template <typename F>
__device__ __forceinline__ void chan_reduce(int n, int R, float *scratch, float *out, F f) {
    const int tid = threadIdx.x;
    const int c = tid % R, sub = tid / R, stride = 256 / R;
    float acc = 0.f;
    if (sub < stride)
        for (int s = sub; s < n; s += stride) acc += f(s, c);
    __syncthreads();
    scratch[tid] = (sub < stride) ? acc : 0.f;
    __syncthreads();
    if (tid < R) {
        float t = 0.f;
        for (int k = 0; k < stride; ++k) t += scratch[k * R + tid];
        out[tid] = t;
    }
    __syncthreads();
}

// Batch mean / rstd of x[n][R] (two-pass) or the moving averages.
__device__ void bn_stats(const float *x, int n, int R, int mode, float eps, float decay, float *m_avg,
                         float *v_avg, float *scratch, float *mean, float *rstd) {
    if (mode == MPNN_ACT_BN_BATCH) {
        chan_reduce(n, R, scratch, mean, [&](int s, int c) { return x[s * R + c]; });
        if (threadIdx.x < R) mean[threadIdx.x] /= (float)n;
        __syncthreads();
        chan_reduce(n, R, scratch, rstd, [&](int s, int c) { const float d = x[s * R + c] - mean[c]; return d * d; });
        if (threadIdx.x < R) {
            const float var = rstd[threadIdx.x] / (float)n;
            m_avg[threadIdx.x] = decay * m_avg[threadIdx.x] + (1.f - decay) * mean[threadIdx.x];
            v_avg[threadIdx.x] = decay * v_avg[threadIdx.x] + (1.f - decay) * var;
            rstd[threadIdx.x] = 1.f / sqrtf(var + eps);
        }
    } else if (threadIdx.x < R) {
        mean[threadIdx.x] = m_avg[threadIdx.x];
        rstd[threadIdx.x] = 1.f / sqrtf(v_avg[threadIdx.x] + eps);
    }
    __syncthreads();
}

__device__ __forceinline__ void head_softmax(const float *z, int n_cls, float *p) {
    float mx = z[0];
    for (int k = 1; k < n_cls; ++k) mx = fmaxf(mx, z[k]);
    float sum = 0.f;
    for (int k = 0; k < n_cls; ++k) { p[k] = expf(z[k] - mx); sum += p[k]; }
    const float inv = 1.f / sum;
    for (int k = 0; k < n_cls; ++k) p[k] *= inv;
}

__global__ __launch_bounds__(256) void exit_tail_fwd_k(const mpnn_exit_tail_args *__restrict__ tab) {
    const mpnn_exit_tail_args &a = tab[blockIdx.x];
    const int tid = threadIdx.x, n = a.n;
    __shared__ float scratch[256];
    __shared__ float bnp[4 * TR];                 // mean1, rstd1, mean2, rstd2
    __shared__ float w2s[TR * TR], w3s[TR * TS], vec[6 * TR + TS];

    if (a.z) {
        for (int s = tid; s < n; s += 256) {
            float p[TC];
            head_softmax(a.z + (size_t)s * a.n_cls, a.n_cls, p);
            const float *y = a.y + (size_t)s * a.n_cls;
            float ce = 0.f; int ap = 0, ay = 0;
            for (int k = 0; k < a.n_cls; ++k) {
                ce -= y[k] * logf(a.eps_ce / (float)a.n_cls + (1.f - a.eps_ce) * p[k]);
                if (p[k] > p[ap]) ap = k;
                if (y[k] > y[ay]) ay = k;
            }
            a.c_err[s] = ce;
            a.d_cor[s] = ap == ay ? 1.f : 0.f;
        }
    }
    if (!a.h1) return;
    const int R = a.R, S = a.n_sinks;
    for (int i = tid; i < R * R; i += 256) w2s[i] = a.w2[i];
    for (int i = tid; i < R * S; i += 256) w3s[i] = a.w3[i];
    if (tid < R) {
        vec[tid] = a.g1[tid]; vec[TR + tid] = a.b1[tid]; vec[2 * TR + tid] = a.bias2[tid];
        vec[3 * TR + tid] = a.g2[tid]; vec[4 * TR + tid] = a.b2[tid];
    }
    if (tid < S) vec[6 * TR + tid] = a.bias3[tid];
    __syncthreads();

    bn_stats(a.h1, n, R, a.mode, a.bn_eps, a.bn_decay, a.m1, a.v1, scratch, bnp, bnp + TR);
    for (int s = tid; s < n; s += 256) {
        float a1[TR];
        for (int c = 0; c < R; ++c)
            a1[c] = fmaxf(vec[c] * (a.h1[s * R + c] - bnp[c]) * bnp[TR + c] + vec[TR + c], 0.f);
        for (int j = 0; j < R; ++j) {
            float h = vec[2 * TR + j];
            for (int c = 0; c < R; ++c) h += a1[c] * w2s[c * R + j];
            a.h2[s * R + j] = h;
        }
    }
    __syncthreads();
    bn_stats(a.h2, n, R, a.mode, a.bn_eps, a.bn_decay, a.m2, a.v2, scratch, bnp + 2 * TR, bnp + 3 * TR);
    for (int s = tid; s < n; s += 256) {
        float a2[TR];
        for (int c = 0; c < R; ++c)
            a2[c] = fmaxf(vec[3 * TR + c] * (a.h2[s * R + c] - bnp[2 * TR + c]) * bnp[3 * TR + c] + vec[4 * TR + c], 0.f);
        for (int i = 0; i < S; ++i) {
            float r = vec[6 * TR + i];
            for (int c = 0; c < R; ++c) r += a2[c] * w3s[c * S + i];
            a.r[(size_t)s * a.r_stride + i] = r;
        }
    }
    if (a.bn_save && tid < R) {
        a.bn_save[tid] = bnp[tid]; a.bn_save[R + tid] = bnp[TR + tid];
        a.bn_save[2 * R + tid] = bnp[2 * TR + tid]; a.bn_save[3 * R + tid] = bnp[3 * TR + tid];
    }
}

extern "C" int mpnn_exit_tail_fwd(const mpnn_exit_tail_args *dev_table, int count, void *stream) {
    if (count <= 0) return 0;
    if (!dev_table) return MPNN_E_ARG;
    hipLaunchKernelGGL(exit_tail_fwd_k, dim3(count), dim3(256), 0, (hipStream_t)stream, dev_table);
    MPNN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------- backward -----------------------------------
// Outer-product sum over the samples staged in LDS: out[i][j] += sum_s A[s][i] * B[s][j].
__device__ __forceinline__ float outer_sum(const float *A, int ai, int as, const float *B, int bj, int bs, int rows) {
    float t = 0.f;
    for (int s = 0; s < rows; ++s) t += A[s * as + ai] * B[s * bs + bj];
    return t;
}

__global__ __launch_bounds__(256) void exit_tail_bwd_k(const mpnn_exit_tail_bwd_args *__restrict__ tab) {
    const mpnn_exit_tail_bwd_args &b = tab[blockIdx.x];
    const mpnn_exit_tail_args &a = b.f;
    const int tid = threadIdx.x, n = a.n;

    if (a.z && b.dz) {
        for (int s = tid; s < n; s += 256) {
            float p[TC], gp[TC];
            head_softmax(a.z + (size_t)s * a.n_cls, a.n_cls, p);
            const float *y = a.y + (size_t)s * a.n_cls;
            const float w = b.w_cerr[s];
            float dot = 0.f;
            for (int k = 0; k < a.n_cls; ++k) {
                const float q = a.eps_ce / (float)a.n_cls + (1.f - a.eps_ce) * p[k];
                gp[k] = -w * y[k] * (1.f - a.eps_ce) / q;
                dot += gp[k] * p[k];
            }
            for (int k = 0; k < a.n_cls; ++k) b.dz[(size_t)s * a.n_cls + k] = p[k] * (gp[k] - dot);
        }
    }
    if (!a.h1) return;

    const int R = a.R, S = a.n_sinks;
    __shared__ float scratch[256];
    __shared__ float w2s[TR * TR], w3s[TR * TS], vec[4 * TR], bnp[4 * TR];
    __shared__ float red[4 * TR];                 // dbeta2, dgamma2, dbeta1, dgamma1
    __shared__ float rowA[CHUNK * TR], rowB[CHUNK * TR];
    for (int i = tid; i < R * R; i += 256) w2s[i] = a.w2[i];
    for (int i = tid; i < R * S; i += 256) w3s[i] = a.w3[i];
    if (tid < R) {
        vec[tid] = a.g1[tid]; vec[TR + tid] = a.b1[tid]; vec[2 * TR + tid] = a.g2[tid]; vec[3 * TR + tid] = a.b2[tid];
        bnp[tid] = a.bn_save[tid]; bnp[TR + tid] = a.bn_save[R + tid];
        bnp[2 * TR + tid] = a.bn_save[2 * R + tid]; bnp[3 * TR + tid] = a.bn_save[3 * R + tid];
    }
    __syncthreads();
    const float inv_n = 1.f / (float)n;

    // Per-sample recomputation helpers (R is tiny; cheaper than round trips).
    auto act1 = [&](int s, float *a1) {
        for (int c = 0; c < R; ++c)
            a1[c] = fmaxf(vec[c] * (a.h1[s * R + c] - bnp[c]) * bnp[TR + c] + vec[TR + c], 0.f);
    };
    auto dh2n_row = [&](int s, float *a2, float *xh2, float *d) {   // grad w.r.t. BN2 output (masked)
        for (int c = 0; c < R; ++c) {
            xh2[c] = (a.h2[s * R + c] - bnp[2 * TR + c]) * bnp[3 * TR + c];
            a2[c] = fmaxf(vec[2 * TR + c] * xh2[c] + vec[3 * TR + c], 0.f);
            float da = 0.f;
            for (int i = 0; i < S; ++i) da += b.dr[(size_t)s * a.r_stride + i] * w3s[c * S + i];
            d[c] = a2[c] > 0.f ? da : 0.f;
        }
    };

    // ---- phase A: dW3, dbias3, dbeta2, dgamma2 --------------------------------
    float accA = 0.f, accB = 0.f, accC = 0.f;      // dW3[tid], dbeta2/dgamma2 partials
    for (int r0 = 0; r0 < n; r0 += CHUNK) {
        const int rows = min(CHUNK, n - r0);
        __syncthreads();
        if (tid < rows) {
            float a2[TR], xh[TR], d[TR];
            dh2n_row(r0 + tid, a2, xh, d);
            for (int c = 0; c < R; ++c) { rowA[tid * TR + c] = a2[c]; rowB[tid * TR + c] = d[c]; }
        }
        __syncthreads();
        if (tid < R * S) {                         // dW3[c][i] += sum a2[s][c] * dr[s][i]
            const int c = tid / S, i = tid - c * S;
            float t = 0.f;
            for (int s = 0; s < rows; ++s) t += rowA[s * TR + c] * b.dr[(size_t)(r0 + s) * a.r_stride + i];
            accA += t;
        }
        if (tid >= 128 && tid < 128 + R) {         // dbeta2, dgamma2
            const int c = tid - 128;
            float t0 = 0.f, t1 = 0.f;
            for (int s = 0; s < rows; ++s) {
                const float d = rowB[s * TR + c];
                const float xh = (a.h2[(r0 + s) * R + c] - bnp[2 * TR + c]) * bnp[3 * TR + c];
                t0 += d; t1 += d * xh;
            }
            accB += t0; accC += t1;
        }
        if (tid >= 192 && tid < 192 + S) {         // dbias3
            const int i = tid - 192;
            float t = 0.f;
            for (int s = 0; s < rows; ++s) t += b.dr[(size_t)(r0 + s) * a.r_stride + i];
            scratch[tid] = (r0 == 0 ? 0.f : scratch[tid]) + t;
        }
    }
    __syncthreads();
    if (tid < R * S) b.dw3[tid] = accA;
    if (tid >= 128 && tid < 128 + R) {
        red[tid - 128] = accB; red[TR + tid - 128] = accC;
        b.db2[tid - 128] = accB; b.dg2[tid - 128] = accC;
    }
    if (tid >= 192 && tid < 192 + S) b.dbias3[tid - 192] = scratch[tid];
    __syncthreads();

    // ---- phase B: dh2 -> dW2, dbias2, then dbeta1, dgamma1 ----------------------
    float accW2 = 0.f, accb2 = 0.f, accD = 0.f, accE = 0.f;
    for (int r0 = 0; r0 < n; r0 += CHUNK) {
        const int rows = min(CHUNK, n - r0);
        __syncthreads();
        if (tid < rows) {
            const int s = r0 + tid;
            float a2[TR], xh[TR], d[TR], a1[TR];
            dh2n_row(s, a2, xh, d);
            act1(s, a1);
            for (int c = 0; c < R; ++c) {
                const float dh2 = vec[2 * TR + c] * bnp[3 * TR + c] * (d[c] - red[c] * inv_n - xh[c] * red[TR + c] * inv_n);
                rowB[tid * TR + c] = dh2;
                rowA[tid * TR + c] = a1[c];
            }
        }
        __syncthreads();
        if (tid < R * R) {                         // dW2[c][j] += sum a1[s][c] * dh2[s][j]
            const int c = tid / R, j = tid - c * R;
            accW2 += outer_sum(rowA, c, TR, rowB, j, TR, rows);
        }
        __syncthreads();
        // dh1n rows (grad w.r.t. BN1 output, masked) overwrite rowA; keep dh2 in rowB.
        if (tid < rows) {
            for (int c = 0; c < R; ++c) {
                float da1 = 0.f;
                for (int j = 0; j < R; ++j) da1 += rowB[tid * TR + j] * w2s[c * R + j];
                rowA[tid * TR + c] = rowA[tid * TR + c] > 0.f ? da1 : 0.f;
            }
        }
        __syncthreads();
        if (tid < R) {                             // dbias2, dbeta1, dgamma1
            float t0 = 0.f, t1 = 0.f, t2 = 0.f;
            for (int s = 0; s < rows; ++s) {
                t0 += rowB[s * TR + tid];
                const float d = rowA[s * TR + tid];
                const float xh = (a.h1[(r0 + s) * R + tid] - bnp[tid]) * bnp[TR + tid];
                t1 += d; t2 += d * xh;
            }
            accb2 += t0; accD += t1; accE += t2;
        }
    }
    __syncthreads();
    if (tid < R * R) b.dw2[tid] = accW2;
    if (tid < R) {
        b.dbias2[tid] = accb2; b.db1[tid] = accD; b.dg1[tid] = accE;
        red[2 * TR + tid] = accD; red[3 * TR + tid] = accE;
    }
    __syncthreads();

    // ---- phase C: dh1 ------------------------------------------------------------
    for (int s = tid; s < n; s += 256) {
        float a2[TR], xh2[TR], d[TR], a1[TR], dh2[TR];
        dh2n_row(s, a2, xh2, d);
        act1(s, a1);
        for (int c = 0; c < R; ++c)
            dh2[c] = vec[2 * TR + c] * bnp[3 * TR + c] * (d[c] - red[c] * inv_n - xh2[c] * red[TR + c] * inv_n);
        for (int c = 0; c < R; ++c) {
            float da1 = 0.f;
            for (int j = 0; j < R; ++j) da1 += dh2[j] * w2s[c * R + j];
            const float dn = a1[c] > 0.f ? da1 : 0.f;
            const float xh1 = (a.h1[s * R + c] - bnp[c]) * bnp[TR + c];
            b.dh1[s * R + c] = vec[c] * bnp[TR + c] * (dn - red[2 * TR + c] * inv_n - xh1 * red[3 * TR + c] * inv_n);
        }
    }
}

extern "C" int mpnn_exit_tail_bwd(const mpnn_exit_tail_bwd_args *dev_table, int count, void *stream) {
    if (count <= 0) return 0;
    if (!dev_table) return MPNN_E_ARG;
    hipLaunchKernelGGL(exit_tail_bwd_k, dim3(count), dim3(256), 0, (hipStream_t)stream, dev_table);
    MPNN_LAUNCH_CHECK();
    return 0;
}
