// Exit tail: everything after the exit's first affine map.
//   head   : Softmax + CrossEntropyError           (layer_types.py:81-84, 262-272)
//   router : BN -> ReLU -> LinTrans(R) -> BN -> ReLU -> LinTrans(n_sinks)
//            (arch_and_hypers.py:47-49; BatchNorm over the batch, layer_types.py:219-239)
// TWO 256-thread workgroups per exit: an even one for the router tail, which
// owns the whole batch (its BatchNorms need statistics over every sample, and
// at R = 16 the arithmetic is a few hundred KFLOP), and an odd one for the
// head (exp / log / divisions of one sample per thread) -- the two halves of
// an exit share nothing, and inside one workgroup the head sat in front of the
// router's serial chain.  The launch is on the step's critical path, so:
// inputs are copied to LDS once (no dependent global round trips between
// phases), every per-thread array has COMPILE-TIME bounds (runtime trip counts
// would push them to scratch memory), reductions go through LDS in a fixed
// order -> deterministic.
// Limits: R <= 16, n_sinks <= 4, n_cls <= 16.  Batches of up to CHUNK = 128 samples run the LDS-resident
// kernels (the shipped specs); larger ones the any-size forms router_fwd_big / router_bwd_big.
#include "common.h"

#define TR 16          // max router width
#define TS 4           // max sinks
#define TC 16          // max classes
#define CHUNK 128      // samples held in LDS
// LDS pitch of a per-sample row of TR floats.  17, not 16: in the per-sample phases lane s reads
// row s, and with a pitch of 16 floats the 64 lanes of a wave fall on 4 banks (16-way conflicts:
// every LDS access of those phases took 16 passes -- 3.6 us per phase in the phase trace).
#define TP 17

// Batch mean / rstd of x[n][TR] or the moving averages.  Channels >= R are inert.
// ONE pass: sums of d = x - p and d^2 around the pivot p = x[0][c], 16 partial sums per channel in a
// fixed order (deterministic), var = E[d^2] - (E[d])^2.  The pivot is a sample of the same distribution,
// so the subtraction loses log2(1 + (mean - p)^2 / var) bits -- a few, of 24; the mean-then-centred
// two-pass form this replaces cost three more barriers and a second sweep per BatchNorm.
// m_old / v_old: the moving averages, loaded by the caller at kernel start (a load here would sit in
// the middle of the serial chain: one more memory round trip per BatchNorm before the barrier).
__device__ void bn_stats(const float *x, int n, int R, int mode, float eps, float decay, float *m_avg,
                         float *v_avg, float m_old, float v_old, float *scratch, float *mean, float *rstd) {
    const int tid = threadIdx.x;
    if (mode == MPNN_ACT_BN_BATCH) {
        const int c = tid & (TR - 1), sub = tid / TR;            // 16 sub-rows of 16 channels
        const float pv = x[c];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < CHUNK / (256 / TR); ++k) {
            const int s = sub + k * (256 / TR);
            const float d = s < n ? x[(s < n ? s : 0) * TP + c] - pv : 0.f;
            s1 += d; s2 += d * d;
        }
        scratch[tid] = s1; scratch[256 + tid] = s2;
        __syncthreads();
        if (tid < TR) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int k = 0; k < 256 / TR; ++k) { t1 += scratch[k * TR + tid]; t2 += scratch[256 + k * TR + tid]; }
            const float inv = 1.f / (float)n, dm = t1 * inv;
            const float mu = pv + dm, var = fmaxf(t2 * inv - dm * dm, 0.f);
            mean[tid] = mu;
            rstd[tid] = rsqrtf(var + eps);
            if (tid < R) {
                m_avg[tid] = decay * m_old + (1.f - decay) * mu;
                v_avg[tid] = decay * v_old + (1.f - decay) * var;
            }
        }
    } else if (tid < R) {
        mean[tid] = m_old;
        rstd[tid] = rsqrtf(v_old + eps);
    }
    __syncthreads();
}

__device__ __forceinline__ void head_softmax(const float *z, int n_cls, float *p) {
    float mx = z[0];
#pragma unroll
    for (int k = 1; k < TC; ++k) if (k < n_cls) mx = fmaxf(mx, z[k]);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < TC; ++k) { p[k] = k < n_cls ? expf(z[k] - mx) : 0.f; sum += p[k]; }
    const float inv = 1.f / sum;
#pragma unroll
    for (int k = 0; k < TC; ++k) p[k] *= inv;
}

// Stage x[n][R] (global, row stride R) into LDS rows of TR, zero-padded: loads and stores are separate
// calls so that a kernel can issue EVERY global load it needs before the first LDS store (a store waits
// for its load: staging array after array was one memory round trip per array, 1-1.5 us each).
// (fixed trip count, loads from clamped addresses issued back to back: a rolled loop with runtime
// bounds made one dependent memory round trip per iteration -- 8 per staged array, 10 us per kernel)
#define ROWS_PT (CHUNK * TR / 256)
__device__ __forceinline__ void load_rows(float (&v)[ROWS_PT], const float *src, int rows, int R) {
#pragma unroll
    for (int k = 0; k < ROWS_PT; ++k) {
        const int i = threadIdx.x + k * 256, s = i / TR, c = i & (TR - 1);
        const bool ok = s < rows && c < R;
        v[k] = src[ok ? s * R + c : 0];
        v[k] = ok ? v[k] : 0.f;
    }
}
__device__ __forceinline__ void store_rows(float *dst, const float (&v)[ROWS_PT], int rows) {
#pragma unroll
    for (int k = 0; k < ROWS_PT; ++k) {
        const int i = threadIdx.x + k * 256, s = i / TR, c = i & (TR - 1);
        if (s < rows) dst[s * TP + c] = v[k];
    }
}


// ---------------------------------------------------------------------------------------------------------
// Router tail for batches of MORE than CHUNK samples (any n; the reference's placeholders are (None, ...),
// net_types.py:50-51).  Same arithmetic, one workgroup per exit, but the batch stays in global memory: every
// thread owns the samples tid, tid + 256, ... in every pass (so it re-reads only rows it wrote itself), the
// batch sums are wave DPP sums + four partials through LDS in a fixed order (deterministic).  Not
// latency-tuned: the hot path of the shipped specs (batch 128) is the kernel above.
// ---------------------------------------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ void block_sums(float (&v)[K], float *part /* [4][K] */, float *tot /* [K] */) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __syncthreads();                                   // (part / tot may still be read from the previous call)
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float w = wave_sum_f(v[k]);
        if (lane == 0) part[wave * K + k] = w;
    }
    __syncthreads();
    for (int k = tid; k < K; k += 256) tot[k] = (part[k] + part[K + k]) + (part[2 * K + k] + part[3 * K + k]);
    __syncthreads();
}

struct RouterW { float *w2s, *w3s, *vec; };            // LDS: W2 [TR][TR], W3 [TR][TS], g1 b1 bias2 g2 b2 [TR each] bias3 [TS]
__device__ __forceinline__ void router_weights(const mpnn_exit_tail_args &a, const RouterW &L) {
    const int tid = threadIdx.x, R = a.R, S = a.n_sinks;
    const int wc = tid / TR, wj = tid & (TR - 1);
    L.w2s[tid] = (wc < R && wj < R) ? a.w2[wc * R + wj] : 0.f;
    if (tid < TR * TS) { const int c = tid / TS, k = tid & (TS - 1); L.w3s[tid] = (c < R && k < S) ? a.w3[c * S + k] : 0.f; }
    if (tid < TR) {
        const bool ok = tid < R;
        L.vec[tid] = ok ? a.g1[tid] : 0.f; L.vec[TR + tid] = ok ? a.b1[tid] : 0.f; L.vec[2 * TR + tid] = ok ? a.bias2[tid] : 0.f;
        L.vec[3 * TR + tid] = ok ? a.g2[tid] : 0.f; L.vec[4 * TR + tid] = ok ? a.b2[tid] : 0.f;
    }
    if (tid < TS) L.vec[5 * TR + tid] = tid < S ? a.bias3[tid] : 0.f;
}
__device__ __forceinline__ void row_load(float (&x)[TR], const float *src, int s, int R) {
#pragma unroll
    for (int c = 0; c < TR; ++c) x[c] = c < R ? src[(size_t)s * R + c] : 0.f;
}
// h2 row of one sample: relu(bn1(h1)) @ W2 + bias2 (a1 returned too)
__device__ __forceinline__ void row_h2(const float (&h1)[TR], const RouterW &L, const float *bnp, float (&a1)[TR], float (&h2)[TR]) {
#pragma unroll
    for (int c = 0; c < TR; ++c) a1[c] = fmaxf(L.vec[c] * (h1[c] - bnp[c]) * bnp[TR + c] + L.vec[TR + c], 0.f);
#pragma unroll
    for (int j = 0; j < TR; ++j) {
        float h = L.vec[2 * TR + j];
#pragma unroll
        for (int c = 0; c < TR; ++c) h += a1[c] * L.w2s[c * TR + j];
        h2[j] = h;
    }
}
// mean / rstd from the pivoted sums (as bn_stats) + the moving-average update
__device__ __forceinline__ void stats_finish(const float *tot, const float *pv, int n, int R, const float bn_eps, const float bn_decay,
                                             float *m_avg, float *v_avg, float *mean, float *rstd) {
    const int tid = threadIdx.x;
    if (tid < TR) {
        const float inv = 1.f / (float)n, dm = tot[tid] * inv;
        const float mu = pv[tid] + dm, var = fmaxf(tot[TR + tid] * inv - dm * dm, 0.f);
        mean[tid] = mu;
        rstd[tid] = rsqrtf(var + bn_eps);
        if (tid < R) {
            m_avg[tid] = bn_decay * m_avg[tid] + (1.f - bn_decay) * mu;
            v_avg[tid] = bn_decay * v_avg[tid] + (1.f - bn_decay) * var;
        }
    }
    __syncthreads();
}

__device__ void router_fwd_big(const mpnn_exit_tail_args &a) {
    __shared__ float w2s[TR * TR], w3s[TR * TS], vec[5 * TR + TS], bnp[4 * TR], pv[TR], part[4 * 2 * TR], tot[2 * TR];
    const RouterW L = {w2s, w3s, vec};
    const int tid = threadIdx.x, n = a.n, R = a.R, S = a.n_sinks;
    router_weights(a, L);
    if (tid < TR) pv[tid] = tid < R ? a.h1[tid] : 0.f;
    if (tid < 4 * TR) bnp[tid] = 0.f;
    __syncthreads();
    const bool batch = a.mode == MPNN_ACT_BN_BATCH;
    float acc[2 * TR];
    if (batch) {
#pragma unroll
        for (int k = 0; k < 2 * TR; ++k) acc[k] = 0.f;
        for (int s = tid; s < n; s += 256) {
            float x[TR];
            row_load(x, a.h1, s, R);
#pragma unroll
            for (int c = 0; c < TR; ++c) { const float d = x[c] - pv[c]; acc[c] += d; acc[TR + c] += d * d; }
        }
        block_sums<2 * TR>(acc, part, tot);
        stats_finish(tot, pv, n, R, a.bn_eps, a.bn_decay, a.m1, a.v1, bnp, bnp + TR);
    } else {
        if (tid < R) { bnp[tid] = a.m1[tid]; bnp[TR + tid] = rsqrtf(a.v1[tid] + a.bn_eps); }
        __syncthreads();
    }
    // pivot of the second BatchNorm: the h2 row of sample 0
    if (tid == 0) {
        float x[TR], a1[TR], h2[TR];
        row_load(x, a.h1, 0, R);
        row_h2(x, L, bnp, a1, h2);
#pragma unroll
        for (int j = 0; j < TR; ++j) pv[j] = h2[j];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2 * TR; ++k) acc[k] = 0.f;
    for (int s = tid; s < n; s += 256) {
        float x[TR], a1[TR], h2[TR];
        row_load(x, a.h1, s, R);
        row_h2(x, L, bnp, a1, h2);
#pragma unroll
        for (int j = 0; j < TR; ++j) {
            if (j < R) a.h2[(size_t)s * R + j] = h2[j];
            const float d = h2[j] - pv[j];
            acc[j] += d; acc[TR + j] += d * d;
        }
    }
    if (batch) {
        block_sums<2 * TR>(acc, part, tot);
        stats_finish(tot, pv, n, R, a.bn_eps2, a.bn_decay2, a.m2, a.v2, bnp + 2 * TR, bnp + 3 * TR);
    } else {
        if (tid < R) { bnp[2 * TR + tid] = a.m2[tid]; bnp[3 * TR + tid] = rsqrtf(a.v2[tid] + a.bn_eps2); }
        __syncthreads();
    }
    for (int s = tid; s < n; s += 256) {
        float h2[TR];
        row_load(h2, a.h2, s, R);                        // (rows this thread wrote itself)
        float a2[TR];
#pragma unroll
        for (int c = 0; c < TR; ++c) a2[c] = fmaxf(vec[3 * TR + c] * (h2[c] - bnp[2 * TR + c]) * bnp[3 * TR + c] + vec[4 * TR + c], 0.f);
#pragma unroll
        for (int i = 0; i < TS; ++i) {
            float r = vec[5 * TR + i];
#pragma unroll
            for (int c = 0; c < TR; ++c) r += a2[c] * w3s[c * TS + i];
            if (i < S) a.r[(size_t)s * a.r_stride + i] = r;
        }
    }
    if (a.bn_save && tid < R) {
        a.bn_save[tid] = bnp[tid]; a.bn_save[R + tid] = bnp[TR + tid];
        a.bn_save[2 * R + tid] = bnp[2 * TR + tid]; a.bn_save[3 * R + tid] = bnp[3 * TR + tid];
    }
}

__global__ __launch_bounds__(256) void exit_tail_fwd_k(const mpnn_exit_tail_args *__restrict__ tab) {
    const mpnn_exit_tail_args a = tab[blockIdx.x >> 1];   // (by value: every field's scalar load in the entry block)
    const int tid = threadIdx.x, n = a.n;
    trace_stamp(0); trace_note(6, 12);
    if (blockIdx.x & 1) {
        // (optional) the accumulators mpnn_route adds to: cleared here, one launch ahead of it
        if (a.clear_f) for (int k = tid; k < a.n_clear_f; k += 256) a.clear_f[k] = 0.f;
        if (a.clear_d) for (int k = tid; k < a.n_clear_d; k += 256) a.clear_d[k] = 0.0;
        if (a.hyp_src && tid < MPNN_HYP_N) a.hyp_dst[tid] = a.hyp_src[tid];     // (this step's schedule values: see the header)
        // ---- the head: Softmax + CrossEntropyError, one sample per thread ----
        if (!a.z) return;
        const int nc = a.n_cls;
        for (int s = tid; s < n; s += 256) {
            float z[TC], y[TC], p[TC];
#pragma unroll
            for (int k = 0; k < TC; ++k) {
                z[k] = k < nc ? a.z[(size_t)s * nc + k] : 0.f;
                y[k] = k < nc ? a.y[(size_t)s * nc + k] : 0.f;
            }
            head_softmax(z, nc, p);
            float ce = 0.f, pmax = p[0], ymax = y[0]; int ap = 0, ay = 0;
#pragma unroll
            for (int k = 0; k < TC; ++k) {
                if (k < nc) {
                    ce -= y[k] * logf(a.eps_ce / (float)nc + (1.f - a.eps_ce) * p[k]);
                    if (p[k] > pmax) { pmax = p[k]; ap = k; }
                    if (y[k] > ymax) { ymax = y[k]; ay = k; }
                }
            }
            a.c_err[s] = ce;
            a.d_cor[s] = ap == ay ? 1.f : 0.f;
        }
        trace_stamp(5);
        return;
    }
    __shared__ float scratch[2 * 256];
    __shared__ float bnp[4 * TR];                 // mean1, rstd1, mean2, rstd2
    __shared__ float w2s[TR * TR], w3s[TR * TS], vec[5 * TR + TS];
    __shared__ float h1s[CHUNK * TP], h2s[CHUNK * TP];

    const bool has_router = a.h1 && n <= CHUNK;    // (larger batches: exit_tail_fwd_big_k)
    if (!has_router) return;
    const int R = a.R, S = a.n_sinks;
    float m1o, v1o, m2o, v2o;
    {
        // every global load of the kernel, then the LDS stores: ONE memory round trip
        float rh[ROWS_PT];
        load_rows(rh, a.h1, n, R);
        const int tc = tid < R ? tid : 0;
        m1o = a.m1[tc]; v1o = a.v1[tc]; m2o = a.m2[tc]; v2o = a.v2[tc];
        const int wc = tid / TR, wj = tid & (TR - 1);
        const bool ok2 = wc < R && wj < R;
        float rw2 = a.w2[ok2 ? wc * R + wj : 0];
        const int w3c = (tid & (TR * TS - 1)) / TS, w3k = tid & (TS - 1);
        const bool ok3 = w3c < R && w3k < S;
        float rw3 = a.w3[ok3 ? w3c * S + w3k : 0];
        const float rg1 = a.g1[tc], rb1 = a.b1[tc], rbias2 = a.bias2[tc], rg2 = a.g2[tc], rb2 = a.b2[tc];
        const float rbias3 = a.bias3[tid < S ? tid : 0];
        store_rows(h1s, rh, n);
        w2s[tid] = ok2 ? rw2 : 0.f;
        if (tid < TR * TS) w3s[tid] = ok3 ? rw3 : 0.f;
        if (tid < TR) {
            const bool ok = tid < R;
            vec[tid] = ok ? rg1 : 0.f; vec[TR + tid] = ok ? rb1 : 0.f; vec[2 * TR + tid] = ok ? rbias2 : 0.f;
            vec[3 * TR + tid] = ok ? rg2 : 0.f; vec[4 * TR + tid] = ok ? rb2 : 0.f;
        }
        if (tid < TS) vec[5 * TR + tid] = tid < S ? rbias3 : 0.f;
        if (tid < 4 * TR) bnp[tid] = 0.f;
    }
    __syncthreads();
    trace_stamp(1);

    bn_stats(h1s, n, R, a.mode, a.bn_eps, a.bn_decay, a.m1, a.v1, m1o, v1o, scratch, bnp, bnp + TR);
    trace_stamp(2);
    {
        // a1 = relu(bn1(h1)); h2 = a1 @ W2 + bias2: two threads per sample, eight output columns each
        const int s = tid >> 1, j0 = (tid & 1) * (TR / 2);
        if (s < n) {
            float a1[TR];
#pragma unroll
            for (int c = 0; c < TR; ++c)
                a1[c] = fmaxf(vec[c] * (h1s[s * TP + c] - bnp[c]) * bnp[TR + c] + vec[TR + c], 0.f);
#pragma unroll
            for (int jj = 0; jj < TR / 2; ++jj) {
                const int j = j0 + jj;
                float h = vec[2 * TR + j];
#pragma unroll
                for (int c = 0; c < TR; ++c) h += a1[c] * w2s[c * TR + j];
                h2s[s * TP + j] = h;
                if (j < R) a.h2[s * R + j] = h;
            }
        }
    }
    __syncthreads();
    trace_stamp(3);
    bn_stats(h2s, n, R, a.mode, a.bn_eps2, a.bn_decay2, a.m2, a.v2, m2o, v2o, scratch, bnp + 2 * TR, bnp + 3 * TR);
    trace_stamp(4);
    for (int s = tid; s < n; s += 256) {
        float a2[TR];
#pragma unroll
        for (int c = 0; c < TR; ++c)
            a2[c] = fmaxf(vec[3 * TR + c] * (h2s[s * TP + c] - bnp[2 * TR + c]) * bnp[3 * TR + c] + vec[4 * TR + c], 0.f);
#pragma unroll
        for (int i = 0; i < TS; ++i) {
            float r = vec[5 * TR + i];
#pragma unroll
            for (int c = 0; c < TR; ++c) r += a2[c] * w3s[c * TS + i];
            if (i < S) a.r[(size_t)s * a.r_stride + i] = r;
        }
    }
    if (a.bn_save && tid < R) {
        a.bn_save[tid] = bnp[tid]; a.bn_save[R + tid] = bnp[TR + tid];
        a.bn_save[2 * R + tid] = bnp[2 * TR + tid]; a.bn_save[3 * R + tid] = bnp[3 * TR + tid];
    }
    trace_stamp(5);
}

int mpnn_trace_install_tail(void *buf) { return mpnn_trace_install(buf); }

// any batch size: one workgroup per exit for the router tail; the heads ride in the LDS-resident kernel's head
// workgroups (they loop over the batch already), launched with the router halves idle
__global__ __launch_bounds__(256) void exit_tail_fwd_big_k(const mpnn_exit_tail_args *__restrict__ tab) {
    const mpnn_exit_tail_args a = tab[blockIdx.x];
    if (a.h1 && a.n > CHUNK) router_fwd_big(a);          // (a record of <= CHUNK samples runs in the LDS-resident kernel only)
}

extern "C" int mpnn_exit_tail_fwd(const mpnn_exit_tail_args *dev_table, int count, int n_max, void *stream) {
    if (count <= 0) return 0;
    if (!dev_table) return MPNN_E_ARG;
    if (n_max > CHUNK) {
        hipLaunchKernelGGL(exit_tail_fwd_big_k, dim3(count), dim3(256), 0, (hipStream_t)stream, dev_table);
        MPNN_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(exit_tail_fwd_k, dim3(2 * count), dim3(256), 0, (hipStream_t)stream, dev_table);
    MPNN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------- backward -----------------------------------
// Sums over the batch as MFMA contractions: D[i][j] = sum_s A[s][i] * B[s][j] over the (<= 128) samples
// held in LDS, 16 x 16 outputs per wave, four samples per v_mfma_f32_16x16x4_f32.  (The first version
// looped over the samples in a handful of threads: four phases of 1.5-5 us each on the critical path.)
// fa(s, i) / fb(s, j): the operands (any expression over LDS rows); rows s >= n contribute zero.
template <typename FA, typename FB>
__device__ __forceinline__ f32x4 contract(int n, FA fa, FB fb) {
    const int lane = threadIdx.x & 63, g = lane >> 4, li = lane & 15;
    // two accumulators (even / odd sample quads): a lone chain of 32 dependent MFMAs is ~36 cycles per link
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int it = 0; it < CHUNK / 4; it += 2) {
        const int s = 4 * it + g, s2 = s + 4;
        const bool ok = s < n, ok2 = s2 < n;
        const float av = ok ? fa(s, li) : 0.f, bv = ok ? fb(s, li) : 0.f;
        const float av2 = ok2 ? fa(s2, li) : 0.f, bv2 = ok2 ? fb(s2, li) : 0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av2, bv2, acc2, 0, 0, 0);
    }
    mfma_drain();
    acc += acc2;
    return acc;                                    // D[row = 4g + r][col = li] = acc[r]
}


// Router tail backward for batches of more than CHUNK samples (see router_fwd_big): every thread owns the samples
// tid, tid + 256, ... and recomputes their short per-sample chains in every pass; the sums over the batch (weight /
// bias gradients, the BatchNorm-backward reductions) are per-thread partial sums + block_sums.
struct RowB { float a2[TR], d2[TR], xh2[TR]; };
__device__ __forceinline__ void row_phase_a(const mpnn_exit_tail_args &a, const float *dr, int s, const RouterW &L, const float *gb /* g1 b1 g2 b2 */,
                                            const float *bnp, float (&drv)[TS], RowB &o) {
    const int R = a.R, S = a.n_sinks;
    float h2[TR];
    row_load(h2, a.h2, s, R);
#pragma unroll
    for (int i = 0; i < TS; ++i) drv[i] = i < S ? dr[(size_t)s * a.r_stride + i] : 0.f;
#pragma unroll
    for (int c = 0; c < TR; ++c) {
        o.xh2[c] = (h2[c] - bnp[2 * TR + c]) * bnp[3 * TR + c];
        o.a2[c] = fmaxf(gb[2 * TR + c] * o.xh2[c] + gb[3 * TR + c], 0.f);
        float da = 0.f;
#pragma unroll
        for (int i = 0; i < TS; ++i) da += drv[i] * L.w3s[c * TS + i];
        o.d2[c] = o.a2[c] > 0.f ? da : 0.f;
    }
}
struct RowC { float a1[TR], dh2[TR], d1[TR], xh1[TR]; };
__device__ __forceinline__ void row_phase_b(const mpnn_exit_tail_args &a, int s, const RouterW &L, const float *gb, const float *bnp,
                                            const float *red, float inv_n, const RowB &rb, RowC &o) {
    const int R = a.R;
    float h1[TR];
    row_load(h1, a.h1, s, R);
#pragma unroll
    for (int c = 0; c < TR; ++c) {
        o.dh2[c] = gb[2 * TR + c] * bnp[3 * TR + c] * (rb.d2[c] - red[c] * inv_n - rb.xh2[c] * red[TR + c] * inv_n);
        o.xh1[c] = (h1[c] - bnp[c]) * bnp[TR + c];
        o.a1[c] = fmaxf(gb[c] * o.xh1[c] + gb[TR + c], 0.f);
    }
#pragma unroll
    for (int c = 0; c < TR; ++c) {
        float da1 = 0.f;
#pragma unroll
        for (int j = 0; j < TR; ++j) da1 += o.dh2[j] * L.w2s[c * TR + j];
        o.d1[c] = o.a1[c] > 0.f ? da1 : 0.f;
    }
}

__device__ void router_bwd_big(const mpnn_exit_tail_bwd_args &b) {
    const mpnn_exit_tail_args &a = b.f;
    constexpr int KA = TR * TS + TS + 2 * TR, KB = 4 * TR + 3 * TR;       // sums of pass A; of one pass-B round
    __shared__ float w2s[TR * TR], w3s[TR * TS], vec[5 * TR + TS], gb[4 * TR], bnp[4 * TR], red[4 * TR];
    constexpr int KM = KA > KB ? KA : KB;
    __shared__ float part[4 * KM], tot[KM];
    const RouterW L = {w2s, w3s, vec};
    const int tid = threadIdx.x, n = a.n, R = a.R, S = a.n_sinks;
    const float inv_n = 1.f / (float)n;
    router_weights(a, L);
    if (tid < TR) {
        const bool ok = tid < R;
        gb[tid] = ok ? a.g1[tid] : 0.f; gb[TR + tid] = ok ? a.b1[tid] : 0.f;
        gb[2 * TR + tid] = ok ? a.g2[tid] : 0.f; gb[3 * TR + tid] = ok ? a.b2[tid] : 0.f;
        bnp[tid] = ok ? a.bn_save[tid] : 0.f; bnp[TR + tid] = ok ? a.bn_save[R + tid] : 0.f;
        bnp[2 * TR + tid] = ok ? a.bn_save[2 * R + tid] : 0.f; bnp[3 * TR + tid] = ok ? a.bn_save[3 * R + tid] : 0.f;
    }
    __syncthreads();
    // ---- pass A: dW3 = a2^T dr, dbias3 = sum dr, dbeta2 = sum d2, dgamma2 = sum d2 * xhat2 ----
    {
        float acc[KA];
#pragma unroll
        for (int k = 0; k < KA; ++k) acc[k] = 0.f;
        for (int s = tid; s < n; s += 256) {
            float drv[TS]; RowB rb;
            row_phase_a(a, b.dr, s, L, gb, bnp, drv, rb);
#pragma unroll
            for (int c = 0; c < TR; ++c) {
#pragma unroll
                for (int i = 0; i < TS; ++i) acc[c * TS + i] += rb.a2[c] * drv[i];
                acc[TR * TS + TS + c] += rb.d2[c];
                acc[TR * TS + TS + TR + c] += rb.d2[c] * rb.xh2[c];
            }
#pragma unroll
            for (int i = 0; i < TS; ++i) acc[TR * TS + i] += drv[i];
        }
        block_sums<KA>(acc, part, tot);
        if (tid < TR * TS) { const int c = tid / TS, i = tid & (TS - 1); if (c < R && i < S) b.dw3[c * S + i] = tot[tid]; }
        if (tid < S) b.dbias3[tid] = tot[TR * TS + tid];
        if (tid < TR) {
            red[tid] = tot[TR * TS + TS + tid]; red[TR + tid] = tot[TR * TS + TS + TR + tid];
            if (tid < R) { b.db2[tid] = red[tid]; b.dg2[tid] = red[TR + tid]; }
        }
        __syncthreads();
    }
    // ---- pass B, four rounds: rows 4q .. 4q+3 of dW2 = a1^T dh2 per round (the first also dbias2 = sum dh2,
    //      dbeta1 = sum d1, dgamma1 = sum d1 * xhat1) ----
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        float acc[KB];
#pragma unroll
        for (int k = 0; k < KB; ++k) acc[k] = 0.f;
        for (int s = tid; s < n; s += 256) {
            float drv[TS]; RowB rb; RowC rc;
            row_phase_a(a, b.dr, s, L, gb, bnp, drv, rb);
            row_phase_b(a, s, L, gb, bnp, red, inv_n, rb, rc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a1r = rc.a1[0];
#pragma unroll
                for (int c = 1; c < TR; ++c) a1r = (c == 4 * q + r) ? rc.a1[c] : a1r;       // (select: no run-time register index)
#pragma unroll
                for (int j = 0; j < TR; ++j) acc[r * TR + j] += a1r * rc.dh2[j];
            }
#pragma unroll
            for (int c = 0; c < TR; ++c) {
                acc[4 * TR + c] += rc.dh2[c];
                acc[5 * TR + c] += rc.d1[c];
                acc[6 * TR + c] += rc.d1[c] * rc.xh1[c];
            }
        }
        block_sums<KB>(acc, part, tot);
        if (tid < 4 * TR) { const int c = 4 * q + tid / TR, j = tid & (TR - 1); if (c < R && j < R) b.dw2[c * R + j] = tot[tid]; }
        if (q == 0 && tid < TR) {
            red[2 * TR + tid] = tot[5 * TR + tid]; red[3 * TR + tid] = tot[6 * TR + tid];
            if (tid < R) { b.dbias2[tid] = tot[4 * TR + tid]; b.db1[tid] = tot[5 * TR + tid]; b.dg1[tid] = tot[6 * TR + tid]; }
        }
        __syncthreads();
    }
    // ---- pass D: dh1 (BatchNorm-1 backward) ----
    for (int s = tid; s < n; s += 256) {
        float drv[TS]; RowB rb; RowC rc;
        row_phase_a(a, b.dr, s, L, gb, bnp, drv, rb);
        row_phase_b(a, s, L, gb, bnp, red, inv_n, rb, rc);
#pragma unroll
        for (int c = 0; c < TR; ++c)
            if (c < R) b.dh1[(size_t)s * R + c] = gb[c] * bnp[TR + c] * (rc.d1[c] - red[2 * TR + c] * inv_n - rc.xh1[c] * red[3 * TR + c] * inv_n);
    }
}

__global__ __launch_bounds__(256) void exit_tail_bwd_k(const mpnn_exit_tail_bwd_args *__restrict__ tab) {
    const mpnn_exit_tail_bwd_args b = tab[blockIdx.x >> 1];   // (by value: every field's scalar load in the entry block)
    const mpnn_exit_tail_args &a = b.f;
    const int tid = threadIdx.x, n = a.n;
    const int lane = tid & 63, g = lane >> 4, li = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    trace_stamp(0); trace_note(6, 13);

    if (blockIdx.x & 1) {
        // ---- the head's backward, one sample per thread ----
        if (!(a.z && b.dz)) return;
        const int nc = a.n_cls;
        for (int hs = tid; hs < n; hs += 256) {
            float hz[TC], hy[TC], p[TC], gp[TC];
#pragma unroll
            for (int k = 0; k < TC; ++k) {
                hz[k] = k < nc ? a.z[(size_t)hs * nc + k] : 0.f;
                hy[k] = k < nc ? a.y[(size_t)hs * nc + k] : 0.f;
            }
            const float hw = b.w_cerr[hs];
            head_softmax(hz, nc, p);
            float dot = 0.f;
#pragma unroll
            for (int k = 0; k < TC; ++k) {
                const float q = a.eps_ce / (float)nc + (1.f - a.eps_ce) * p[k];
                gp[k] = k < nc ? -hw * hy[k] * (1.f - a.eps_ce) / q : 0.f;
                dot += gp[k] * p[k];
            }
#pragma unroll
            for (int k = 0; k < TC; ++k) if (k < nc) b.dz[(size_t)hs * nc + k] = p[k] * (gp[k] - dot);
        }
        trace_stamp(5);
        return;
    }
    const bool has_router = a.h1 && n <= CHUNK;    // (larger batches: exit_tail_bwd_big_k)
    if (!has_router) return;

    const int R = a.R, S = a.n_sinks;
    __shared__ float w2s[TR * TR], w3s[TR * TS], vec[4 * TR], bnp[4 * TR];
    __shared__ float red[4 * TR];                 // dbeta2, dgamma2, dbeta1, dgamma1
    __shared__ float rowA[CHUNK * TP], rowB[CHUNK * TP], rowC[CHUNK * TP];
    __shared__ float h1s[CHUNK * TP], h2s[CHUNK * TP], drs[CHUNK * TS];
    {
        // every global load of the kernel, then the LDS stores: ONE memory round trip
        float rh1[ROWS_PT], rh2[ROWS_PT], rdr[CHUNK * TS / 256];
        load_rows(rh1, a.h1, n, R);
        load_rows(rh2, a.h2, n, R);
#pragma unroll
        for (int kk = 0; kk < CHUNK * TS / 256; ++kk) {
            const int i = tid + kk * 256, s = i / TS, k = i & (TS - 1);
            const bool ok = s < n && k < S;
            rdr[kk] = b.dr[ok ? (size_t)s * a.r_stride + k : 0];
            rdr[kk] = ok ? rdr[kk] : 0.f;
        }
        const int wc = tid / TR, wj = tid & (TR - 1);
        const bool ok2 = wc < R && wj < R;
        const float rw2 = a.w2[ok2 ? wc * R + wj : 0];
        const int w3c = (tid & (TR * TS - 1)) / TS, w3k = tid & (TS - 1);
        const bool ok3 = w3c < R && w3k < S;
        const float rw3 = a.w3[ok3 ? w3c * S + w3k : 0];
        const int tc = tid < R ? tid : 0;
        const float rg1 = a.g1[tc], rb1 = a.b1[tc], rg2 = a.g2[tc], rb2 = a.b2[tc];
        const float s0 = a.bn_save[tc], s1 = a.bn_save[R + tc], s2 = a.bn_save[2 * R + tc], s3 = a.bn_save[3 * R + tc];
        store_rows(h1s, rh1, n);
        store_rows(h2s, rh2, n);
#pragma unroll
        for (int kk = 0; kk < CHUNK * TS / 256; ++kk) {
            const int i = tid + kk * 256;
            if (i < n * TS) drs[i] = rdr[kk];
        }
        w2s[tid] = ok2 ? rw2 : 0.f;
        if (tid < TR * TS) w3s[tid] = ok3 ? rw3 : 0.f;
        if (tid < TR) {
            const bool ok = tid < R;
            vec[tid] = ok ? rg1 : 0.f; vec[TR + tid] = ok ? rb1 : 0.f;
            vec[2 * TR + tid] = ok ? rg2 : 0.f; vec[3 * TR + tid] = ok ? rb2 : 0.f;
            bnp[tid] = ok ? s0 : 0.f; bnp[TR + tid] = ok ? s1 : 0.f;
            bnp[2 * TR + tid] = ok ? s2 : 0.f; bnp[3 * TR + tid] = ok ? s3 : 0.f;
        }
    }
    __syncthreads();
    trace_stamp(1);
    const float inv_n = 1.f / (float)n;
    auto xh2 = [&](int s, int c) { return (h2s[s * TP + c] - bnp[2 * TR + c]) * bnp[3 * TR + c]; };
    auto xh1 = [&](int s, int c) { return (h1s[s * TP + c] - bnp[c]) * bnp[TR + c]; };
    // the per-sample phases: two threads per sample, eight channels each
    const int ps = tid >> 1, pc0 = (tid & 1) * (TR / 2);

    // ---- phase A: per-sample a2 and the masked dL/d(bn2 out) ----
    if (ps < n) {
        float drv[TS];
#pragma unroll
        for (int i = 0; i < TS; ++i) drv[i] = drs[ps * TS + i];
#pragma unroll
        for (int cc = 0; cc < TR / 2; ++cc) {
            const int c = pc0 + cc;
            const float a2 = fmaxf(vec[2 * TR + c] * xh2(ps, c) + vec[3 * TR + c], 0.f);
            float da = 0.f;
#pragma unroll
            for (int i = 0; i < TS; ++i) da += drv[i] * w3s[c * TS + i];
            rowA[ps * TP + c] = a2;
            rowB[ps * TP + c] = a2 > 0.f ? da : 0.f;
        }
    }
    __syncthreads();
    // dW3 = a2^T dr | dbias3 = sum dr | dbeta2 = sum d, dgamma2 = sum d * xhat2: one wave each
    if (wave == 0) {
        const f32x4 d = contract(n, [&](int s, int i) { return rowA[s * TP + i]; },
                                 [&](int s, int j) { return j < TS ? drs[s * TS + j] : 0.f; });
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int c = 4 * g + r; if (c < R && li < S) b.dw3[c * S + li] = d[r]; }
    } else if (wave == 1) {
        const f32x4 d = contract(n, [&](int, int i) { return i == 0 ? 1.f : 0.f; },
                                 [&](int s, int j) { return j < TS ? drs[s * TS + j] : 0.f; });
        if (g == 0 && li < S) b.dbias3[li] = d[0];
    } else if (wave == 2) {
        const f32x4 d = contract(n, [&](int, int i) { return i == 0 ? 1.f : 0.f; },
                                 [&](int s, int j) { return rowB[s * TP + j]; });
        if (g == 0) { red[li] = d[0]; if (li < R) b.db2[li] = d[0]; }
    } else {
        const f32x4 d = contract(n, [&](int, int i) { return i == 0 ? 1.f : 0.f; },
                                 [&](int s, int j) { return rowB[s * TP + j] * xh2(s, j); });
        if (g == 0) { red[TR + li] = d[0]; if (li < R) b.dg2[li] = d[0]; }
    }
    __syncthreads();
    trace_stamp(2);

    // ---- phase B: dh2 (BatchNorm-2 backward) and a1 ----
    if (ps < n) {
#pragma unroll
        for (int cc = 0; cc < TR / 2; ++cc) {
            const int c = pc0 + cc;
            const float dh2 = vec[2 * TR + c] * bnp[3 * TR + c] * (rowB[ps * TP + c] - red[c] * inv_n - xh2(ps, c) * red[TR + c] * inv_n);
            rowB[ps * TP + c] = dh2;
            rowA[ps * TP + c] = fmaxf(vec[c] * xh1(ps, c) + vec[TR + c], 0.f);   // a1
        }
    }
    __syncthreads();
    // per-sample masked dL/d(bn1 out) -> rowC (needs the sample's whole dh2 row: hence the barrier)
    if (ps < n) {
        float dh2[TR];
#pragma unroll
        for (int j = 0; j < TR; ++j) dh2[j] = rowB[ps * TP + j];
#pragma unroll
        for (int cc = 0; cc < TR / 2; ++cc) {
            const int c = pc0 + cc;
            float da1 = 0.f;
#pragma unroll
            for (int j = 0; j < TR; ++j) da1 += dh2[j] * w2s[c * TR + j];
            rowC[ps * TP + c] = rowA[ps * TP + c] > 0.f ? da1 : 0.f;
        }
    }
    __syncthreads();
    trace_stamp(3);
    // dW2 = a1^T dh2 | dbias2 = sum dh2 | dbeta1 = sum d1 | dgamma1 = sum d1 * xhat1: one wave each
    if (wave == 0) {
        const f32x4 d = contract(n, [&](int s, int i) { return rowA[s * TP + i]; }, [&](int s, int j) { return rowB[s * TP + j]; });
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int c = 4 * g + r; if (c < R && li < R) b.dw2[c * R + li] = d[r]; }
    } else if (wave == 1) {
        const f32x4 d = contract(n, [&](int, int i) { return i == 0 ? 1.f : 0.f; }, [&](int s, int j) { return rowB[s * TP + j]; });
        if (g == 0 && li < R) b.dbias2[li] = d[0];
    } else if (wave == 2) {
        const f32x4 d = contract(n, [&](int, int i) { return i == 0 ? 1.f : 0.f; }, [&](int s, int j) { return rowC[s * TP + j]; });
        if (g == 0) { red[2 * TR + li] = d[0]; if (li < R) b.db1[li] = d[0]; }
    } else {
        const f32x4 d = contract(n, [&](int, int i) { return i == 0 ? 1.f : 0.f; }, [&](int s, int j) { return rowC[s * TP + j] * xh1(s, j); });
        if (g == 0) { red[3 * TR + li] = d[0]; if (li < R) b.dg1[li] = d[0]; }
    }
    __syncthreads();
    trace_stamp(4);
    // ---- phase D: dh1 (BatchNorm-1 backward) ----
    for (int i = tid; i < n * TR; i += 256) {
        const int s = i / TR, c = i & (TR - 1);
        if (c < R)
            b.dh1[s * R + c] = vec[c] * bnp[TR + c] * (rowC[s * TP + c] - red[2 * TR + c] * inv_n - xh1(s, c) * red[3 * TR + c] * inv_n);
    }
    trace_stamp(5);
}

__global__ __launch_bounds__(256) void exit_tail_bwd_big_k(const mpnn_exit_tail_bwd_args *__restrict__ tab) {
    const mpnn_exit_tail_bwd_args b = tab[blockIdx.x];
    if (b.f.h1 && b.f.n > CHUNK) router_bwd_big(b);
}

extern "C" int mpnn_exit_tail_bwd(const mpnn_exit_tail_bwd_args *dev_table, int count, int n_max, void *stream) {
    if (count <= 0) return 0;
    if (!dev_table) return MPNN_E_ARG;
    if (n_max > CHUNK) {
        hipLaunchKernelGGL(exit_tail_bwd_big_k, dim3(count), dim3(256), 0, (hipStream_t)stream, dev_table);
        MPNN_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(exit_tail_bwd_k, dim3(2 * count), dim3(256), 0, (hipStream_t)stream, dev_table);
    MPNN_LAUNCH_CHECK();
    return 0;
}
