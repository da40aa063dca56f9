// mpnn_exit_ev: one exit of the routing tree in EVALUATION mode, fused, for any batch size.
//
//   head   : Select(-1) -> LinTrans -> Softmax -> CrossEntropyError      (arch_and_hypers.py:66-70,
//            layer_types.py:39-53, 81-84, 262-272)
//   router : Select(-1) -> LinTrans(R) -> BN -> ReLU -> LinTrans(R) -> BN -> ReLU -> LinTrans(n_sinks)
//            (arch_and_hypers.py:45-49) with MOVING-AVERAGE BatchNorm (layer_types.py:237-238)
//   routing: pi_ev = one-hot(arg-max r), first index on ties (net_types.py:127-129)
//
// In 'ev' mode every sample is independent (no batch statistics), so a workgroup owns 16 samples
// end to end: the two affine maps over the block's coarsest scale (BatchNorm + ReLU applied on
// load) run on v_mfma_f32_16x16x4_f32 with K split over 16 waves, the partial tiles meet in LDS,
// and 16 threads finish one sample each.  No 128-sample cap (mpnn_exit_tail_fwd needs the whole
// batch in one workgroup for its batch statistics; this kernel does not).
//
// Routed evaluation: the launch works on the sample list idx[0..*cnt) of its node (device-side
// count) and APPENDS every sample to the list of the child it is routed to: wave64 ballot of
// `arg-max == sink`, popcount prefix for the rank inside the wave, ONE atomicAdd per (wave, sink)
// to reserve the range.  Nothing returns to the host; the child's launches read the count on the
// device.  The order of a list depends on workgroup timing; results do not (every sample's
// arithmetic is independent of its slot).
#include "common.h"

#define EV_WAVES 16
#define TR 16          // max router width
#define TS MPNN_MAX_SINKS
#define TC 16          // max classes

__global__ __launch_bounds__(EV_WAVES * 64) void exit_ev_k(const mpnn_exit_ev_args *__restrict__ tab) {
    const mpnn_exit_ev_args &a = tab[blockIdx.y];
    int n = a.n;
    if (a.cnt) { const int c = *a.cnt; n = c < n ? c : n; }
    const int n0 = blockIdx.x * 16;
    if (n0 >= n) return;
    __shared__ float cA[128 * 3];
    __shared__ float red[EV_WAVES * 2 * 256];
    __shared__ float zs[16 * TC], hs[16 * TR];
    __shared__ float w2s[TR * TR], w3s[TR * TS], vec[9 * TR + TS];
    __shared__ int img_s[16];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = a.a.C, K = a.HW * C;
    const bool bn = a.a.mode != MPNN_ACT_IDENTITY;
    const bool has_head = a.w_head != nullptr, has_router = a.w1 != nullptr;
    const int R = has_router ? a.R : 0, S = has_router ? a.n_sinks : 0;
    if (bn) {
        for (int c = tid; c < C; c += EV_WAVES * 64) {
            const BnC k = bn_coef(a.a, c);
            cA[c * 3] = k.m; cA[c * 3 + 1] = k.gamma * k.rstd; cA[c * 3 + 2] = k.beta;
        }
    }
    if (tid < 16) img_s[tid] = n0 + tid < n ? (a.idx ? a.idx[n0 + tid] : n0 + tid) : -1;
    // router tail parameters -> LDS (requested now, used after the affine maps)
    if (has_router) {
        if (tid < TR * TR) { const int c = tid / TR, j = tid & (TR - 1); w2s[tid] = (c < R && j < R) ? a.w2[c * R + j] : 0.f; }
        else if (tid < TR * TR + TR * TS) {
            const int i = tid - TR * TR, c = i / TS, k = i % TS;
            w3s[i] = (c < R && k < S) ? a.w3[c * S + k] : 0.f;
        } else if (tid < TR * TR + TR * TS + TR) {
            const int c = tid - TR * TR - TR * TS;
            const bool ok = c < R;
            // BatchNorm with moving averages folded to scale/shift: y = k*(x - m) + beta
            const float k1 = ok ? a.g1[c] * rsqrtf(a.v1[c] + a.bn_eps) : 0.f, k2 = ok ? a.g2[c] * rsqrtf(a.v2[c] + a.bn_eps2) : 0.f;
            vec[c] = k1; vec[TR + c] = ok ? a.m1[c] : 0.f; vec[2 * TR + c] = ok ? a.be1[c] : 0.f;
            vec[3 * TR + c] = k2; vec[4 * TR + c] = ok ? a.m2[c] : 0.f; vec[5 * TR + c] = ok ? a.be2[c] : 0.f;
            vec[6 * TR + c] = ok ? a.bias2[c] : 0.f;
            vec[7 * TR + c] = ok ? a.b1[c] : 0.f;
            vec[8 * TR + c] = (ok && a.extra_col) ? a.w1[(size_t)K * R + c] : 0.f;
            if (c < TS) vec[9 * TR + c] = c < S ? a.bias3[c] : 0.f;
        }
    }
    __syncthreads();
    const int img = img_s[li];
    const bool valid = img >= 0;
    const int M0 = has_head ? a.n_cls : 0, M1 = R;
    const float *xrow = a.a.x + (size_t)(valid ? img : 0) * K;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    // two 16-feature blocks per iteration: their loads are in flight together
    for (int kb = wid * 2; kb < (K >> 4); kb += EV_WAVES * 2) {
        f32x4 x[2];
        float b0[2][4], b1[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int kk = kb + u < (K >> 4) ? kb + u : kb;           // (K/16 odd: the second block repeats the first, masked below)
            const int k = kk * 16 + 4 * g;
            x[u] = *(const f32x4 *)(xrow + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                b0[u][j] = li < M0 ? a.w_head[(size_t)(k + j) * M0 + li] : 0.f;
                b1[u][j] = li < M1 ? a.w1[(size_t)(k + j) * M1 + li] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bool on = valid && kb + u < (K >> 4);
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
    mfma_drain();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[(wid * 2 + 0) * 256 + lane * 4 + r] = acc0[r];
        red[(wid * 2 + 1) * 256 + lane * 4 + r] = acc1[r];
    }
    __syncthreads();
    if (tid < 512) {                         // (set, lane, r): D row = 4*(lane>>4) + r (sample), col = lane & 15 (output)
        const int s = tid >> 8, e = tid & 255, l = e >> 2, r = e & 3;
        const int row = (l >> 4) * 4 + r, col = l & 15;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < EV_WAVES; ++w) v += red[(w * 2 + s) * 256 + e];
        if (s == 0) zs[row * TC + col] = (has_head && col < M0) ? v + a.b_head[col] : 0.f;
        else hs[row * TR + col] = col < M1 ? v : 0.f;          // (bias and k_cpt column: in the tail, from LDS)
    }
    __syncthreads();
    if (tid >= 256) return;                  // waves 0-3 stay (a barrier counts the waves that are still alive)
    // ---- router tail on 256 threads: thread (sample s, column c) computes ONE output of each of the three maps ----
    // (one sample per lane did all 16 x 16 + 16 x S multiply-adds of a sample in sequence: ~2.5 us of a lone wave)
    __shared__ float a1s[16 * TR], a2s[16 * TR], rs[16 * TS];
    __shared__ int arg_s[16];
    if (has_router) {
        const int sm = tid >> 4, c = tid & 15;
        const int my_s = img_s[sm];
        const float kc = (a.extra_col && my_s >= 0) ? a.alpha_cpt * a.k_cpt[my_s] : 0.f;
        const float h1 = hs[sm * TR + c] + vec[7 * TR + c] + kc * vec[8 * TR + c];
        a1s[tid] = fmaxf(vec[c] * (h1 - vec[TR + c]) + vec[2 * TR + c], 0.f);
        __syncthreads();
        float h = vec[6 * TR + c];
#pragma unroll
        for (int cc = 0; cc < TR; ++cc) h += a1s[sm * TR + cc] * w2s[cc * TR + c];
        a2s[tid] = fmaxf(vec[3 * TR + c] * (h - vec[4 * TR + c]) + vec[5 * TR + c], 0.f);
        __syncthreads();
        if (c < TS) {
            float r = vec[9 * TR + c];
#pragma unroll
            for (int cc = 0; cc < TR; ++cc) r += a2s[sm * TR + cc] * w3s[cc * TS + c];
            rs[sm * TS + c] = r;
            if (c < S && my_s >= 0) a.r[(size_t)my_s * a.r_stride + c] = r;
        }
        __syncthreads();
        if (c == 0) {
            int arg = 0; float rmax = rs[sm * TS];
#pragma unroll
            for (int i = 1; i < TS; ++i) if (i < S && rs[sm * TS + i] > rmax) { rmax = rs[sm * TS + i]; arg = i; }   // first index on ties (tf.argmax)
            arg_s[sm] = arg;
        }
        __syncthreads();
    }
    // ---- wave 0: the head (one sample per lane); wave 1, at the same time: the children's lists ----
    if (wid == 0) {
        const bool mine = tid < 16 && img_s[tid < 16 ? tid : 0] >= 0;
        const int my = mine ? img_s[tid] : 0;
        if (has_head && mine) {
            const int nc = a.n_cls;
            float z[TC], p[TC];
#pragma unroll
            for (int k = 0; k < TC; ++k) z[k] = zs[tid * TC + k];
            float mx = z[0];
#pragma unroll
            for (int k = 1; k < TC; ++k) if (k < nc) mx = fmaxf(mx, z[k]);
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < TC; ++k) { p[k] = k < nc ? expf(z[k] - mx) : 0.f; sum += p[k]; }
            const float inv = 1.f / sum;
            float ce = 0.f, pmax = 0.f, ymax = 0.f; int ap = 0, ay = 0;
#pragma unroll
            for (int k = 0; k < TC; ++k) {
                if (k < nc) {
                    const float pk = p[k] * inv, yk = a.y[(size_t)my * nc + k];
                    ce -= yk * logf(a.eps_ce / (float)nc + (1.f - a.eps_ce) * pk);
                    if (k == 0 || pk > pmax) { pmax = pk; ap = k; }
                    if (k == 0 || yk > ymax) { ymax = yk; ay = k; }
                }
            }
            a.c_err[my] = ce;
            a.d_cor[my] = ap == ay ? 1.f : 0.f;
        }
        return;
    }
    if (wid != 1 || !has_router) return;
    // ---- compaction into the children's lists: ballot + popcount prefix, one atomic per (wave, sink); the atomics
    // of all sinks are issued before the first result is used (they are independent round trips) ----
    const bool mine = lane < 16 && img_s[lane < 16 ? lane : 0] >= 0;
    const int my = mine ? img_s[lane] : 0;
    const int arg = mine ? arg_s[lane] : 0;
    unsigned long long m[TS];
    int base[TS];
#pragma unroll
    for (int i = 0; i < TS; ++i) {
        m[i] = (i < S && a.child_idx[i]) ? __ballot(mine && arg == i) : 0ull;          // (uniform condition)
        base[i] = 0;
        if (m[i] && lane == 0) base[i] = atomicAdd(a.child_cnt[i], (int)__popcll(m[i]));
    }
#pragma unroll
    for (int i = 0; i < TS; ++i) {
        if (!m[i]) continue;
        const int b0 = __shfl(base[i], 0);
        const int pos = b0 + __popcll(m[i] & ((1ull << lane) - 1ull));
        if (mine && arg == i && pos < a.n) a.child_idx[i][pos] = my;       // (a list holds at most n samples: counts not cleared by the caller must not write past it)
    }
}

extern "C" int mpnn_exit_ev(const mpnn_exit_ev_args *dev_table, int count, int n_max, void *stream) {
    if (count <= 0 || n_max <= 0) return 0;
    if (!dev_table) return MPNN_E_ARG;
    hipLaunchKernelGGL(exit_ev_k, dim3((n_max + 15) / 16, count), dim3(EV_WAVES * 64), 0, (hipStream_t)stream, dev_table);
    MPNN_LAUNCH_CHECK();
    return 0;
}

// Host-side check of one record against the kernel's compile-time limits (the table itself lives in
// device memory, so the caller validates each record before uploading it).
extern "C" int mpnn_exit_ev_check(const mpnn_exit_ev_args *host_rec) {
    if (!host_rec || !host_rec->a.x) return MPNN_E_ARG;
    const mpnn_exit_ev_args &a = *host_rec;
    if (a.a.C > 128 || a.a.C <= 0 || (a.a.C & 3) || ((a.HW * a.a.C) & 15)) return MPNN_E_SHAPE;
    if (a.w_head && (a.n_cls < 1 || a.n_cls > TC || !a.b_head || !a.y || !a.c_err || !a.d_cor)) return a.n_cls > TC ? MPNN_E_SHAPE : MPNN_E_ARG;
    if (a.w1) {
        if (a.R < 1 || a.R > TR || a.n_sinks < 2 || a.n_sinks > TS) return MPNN_E_SHAPE;
        if (!a.b1 || !a.g1 || !a.be1 || !a.m1 || !a.v1 || !a.w2 || !a.bias2 || !a.g2 || !a.be2 || !a.m2 || !a.v2 ||
            !a.w3 || !a.bias3 || !a.r || a.r_stride < a.n_sinks) return MPNN_E_ARG;
        if (a.extra_col && !a.k_cpt) return MPNN_E_ARG;
        for (int i = 0; i < TS; ++i) if ((a.child_idx[i] != nullptr) != (a.child_cnt[i] != nullptr)) return MPNN_E_ARG;
    }
    if (a.idx && !a.cnt) return MPNN_E_ARG;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// mpnn_ev_prefix_walk: the routed evaluation's DENSE PREFIX made routed after the fact.
//
// A routed pass is a chain of (conv, exit) launches per tree depth: a router's decision gates the NEXT exit, so eight
// blocks are eight serial exit launches of ~15 us where the dense pass has one -- which is what keeps the routed pass
// slower than the dense one below ~1 000 samples, where running deep blocks on fewer samples saves nothing.  The blocks
// of the first d0 depths run their convs on every sample anyway (Engine.routed_prefix); so their exits run densely too,
// in ONE mpnn_exit_ev launch, and this kernel restores the routed pass's contract afterwards: one thread per sample
// walks the prefix's switches top-down (arg-max of the router outputs, first index on ties -- the rule of exit_ev_k),
// ZEROES r / c_err / delta_cor of every prefix node the sample does not reach, and appends the sample to the list of the
// frontier block (depth d0) it arrives at (ballot + popcount prefix, one atomic per wave and list, as in exit_ev_k).
// Results are identical to the exit-by-exit routed pass: same kernels, same per-sample arithmetic, same lists.
__global__ __launch_bounds__(256) void ev_prefix_k(const mpnn_ev_prefix_args *__restrict__ tab) {
    const mpnn_ev_prefix_args &a = *tab;
    const int s = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    const bool live = s < a.n;
    const size_t sv = live ? s : 0;
    // pass 1: every switch's decision (2 bits each); the loads of different records are independent
    unsigned long long arg_lo = 0ull, arg_hi = 0ull;
#pragma unroll 4
    for (int j = 0; j < a.count; ++j) {
        const int S = a.n_sinks[j];
        if (S <= 0) continue;
        const float *r = a.r[j] + sv * a.r_stride[j];
        int arg = 0; float best = r[0];
        for (int i = 1; i < S; ++i) { const float v = r[i]; if (v > best) { best = v; arg = i; } }
        if (j < 32) arg_lo |= (unsigned long long)arg << (2 * j); else arg_hi |= (unsigned long long)arg << (2 * (j - 32));
    }
    auto arg_of = [&](int j) { return (int)(((j < 32 ? arg_lo >> (2 * j) : arg_hi >> (2 * (j - 32)))) & 3ull); };
    // pass 2: reach bits in topological order (parents first), entries of unreached nodes cleared
    unsigned long long reach = 0ull;
    for (int j = 0; j < a.count; ++j) {
        const int p = a.parent[j];
        const bool here = p < 0 || (((reach >> p) & 1ull) && arg_of(p) == a.parent_sink[j]);
        if (here) reach |= 1ull << j;
        else if (live) {
            if (a.c_err[j]) { a.c_err[j][s] = 0.f; a.d_cor[j][s] = 0.f; }
            if (a.n_sinks[j] > 0) { float *r = a.r[j] + sv * a.r_stride[j]; for (int i = 0; i < a.n_sinks[j]; ++i) r[i] = 0.f; }
        }
    }
    // pass 3: the frontier blocks' sample lists
    for (int f = 0; f < a.n_front; ++f) {
        const int p = a.front_parent[f];
        const bool mine = live && ((reach >> p) & 1ull) && arg_of(p) == a.front_sink[f];
        const unsigned long long m = __ballot(mine);
        if (!m) continue;                                   // (wave-uniform)
        int base = 0;
        if (lane == 0) base = atomicAdd(a.front_cnt[f], (int)__popcll(m));
        base = __shfl(base, 0);
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        if (mine && pos < a.n) a.front_idx[f][pos] = s;
    }
}

extern "C" int mpnn_ev_prefix_walk(const mpnn_ev_prefix_args *host_rec, const mpnn_ev_prefix_args *dev_rec, void *stream) {
    if (!host_rec || !dev_rec) return MPNN_E_ARG;
    const mpnn_ev_prefix_args &a = *host_rec;
    if (a.n <= 0) return 0;
    if (a.count < 1 || a.count > MPNN_PREFIX_MAX || a.n_front < 0 || a.n_front > MPNN_PREFIX_MAX) return MPNN_E_SHAPE;
    for (int j = 0; j < a.count; ++j) {
        if (a.parent[j] >= j) return MPNN_E_ARG;                                  // (parents first)
        if (a.parent[j] >= 0 && (a.n_sinks[a.parent[j]] <= 0 || a.parent_sink[j] < 0 || a.parent_sink[j] >= a.n_sinks[a.parent[j]])) return MPNN_E_ARG;
        if (a.n_sinks[j] < 0 || a.n_sinks[j] > MPNN_MAX_SINKS) return MPNN_E_SHAPE;
        if (a.n_sinks[j] > 0 && (!a.r[j] || a.r_stride[j] < a.n_sinks[j])) return MPNN_E_ARG;
        if ((a.c_err[j] != nullptr) != (a.d_cor[j] != nullptr)) return MPNN_E_ARG;
    }
    for (int f = 0; f < a.n_front; ++f) {
        const int p = a.front_parent[f];
        if (p < 0 || p >= a.count || a.n_sinks[p] <= 0 || a.front_sink[f] < 0 || a.front_sink[f] >= a.n_sinks[p] ||
            !a.front_idx[f] || !a.front_cnt[f]) return MPNN_E_ARG;
    }
    hipLaunchKernelGGL(ev_prefix_k, dim3((a.n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dev_rec);
    MPNN_LAUNCH_CHECK();
    return 0;
}
