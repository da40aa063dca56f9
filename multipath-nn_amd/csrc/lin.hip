// Exit-path affine maps: the LogReg head's LinTrans and the router's first
// LinTrans share one pass over the block's coarsest scale (BatchNorm + ReLU
// applied on load).  These launches sit on the step's critical path (forward
// trunk -> exits -> router -> exits backward -> trunk backward) with almost no
// arithmetic, so they are organised for latency: table-driven (one launch for
// every exit of the tree), K split over the waves of a workgroup (and, for the
// small batches of the training step, over workgroups) in the forward, and ONE
// backward kernel that produces dX, dW and db (and the fused BatchNorm-backward
// reductions) from one load of X on MFMA tiles.  Where workgroups share an
// output they meet through a ticket and a fixed-order sum: no float atomics.
#include "common.h"

// ------------------------------- forward ------------------------------------
// 16 samples x 16 outputs per MFMA tile; waves split K, partial tiles meet in LDS.
// LFW waves per workgroup.  16 (mpnn_lin_fwd): one workgroup per 16 rows owns all of K.
// 4 (mpnn_lin_fwd_ks, small batches): a record with K >= 512 is split over S = min(8, K / 256)
// workgroups per 16 rows -- the K = 1024 workgroups of the unsliced form pull 170 KB (weights re-read
// by every row group + activations) through ONE compute unit, 8 us of a 14 us launch.  A slice leaves
// its partial tile in scratch with write-through (system-scope) stores and takes a ticket; the last to
// arrive adds the S partials in slice order, so the result does not depend on who that is.  No
// agent-scope fence anywhere: on this part a release fence writes back the XCD's whole L2 (measured:
// the launch took 83 us with __threadfence()).  Four waves, not sixteen: the dispatcher starts
// 240 sixteen-wave workgroups over 7 us.
template <int LFW, bool SLICED>
__global__ __launch_bounds__(LFW * 64) void lin_fwd_k(const mpnn_lin_fwd_args *__restrict__ tab, const int n_rec) {
    // (by value: every field's scalar load sits in the entry block, one round trip.  The two-element arrays are
    // only ever indexed by CONSTANTS below -- a run-time index would put the copy in scratch memory: +4 us)
    const mpnn_lin_fwd_args a = tab[SLICED ? blockIdx.y % n_rec : blockIdx.y];     // (sliced: y = slice * n_rec + record)
    const int n0 = blockIdx.x * 16;
    if (n0 >= a.n) return;
    const int slice = SLICED ? blockIdx.y / n_rec : 0;
    const int S = (SLICED && a.kpart && a.kcnt) ? min(MPNN_LIN_KSLICES, max(1, (a.HW * a.a.C) >> 8)) : 1;
    if (slice >= S) return;
    trace_stamp(0); trace_note(6, 10);
    constexpr int NT_ = LFW * 64, NO = 512 / (NT_ < 512 ? NT_ : 512);        // outputs per thread: (set, lane, r) = 512
    __shared__ float cA[128 * 3];
    __shared__ float red[LFW * 2 * 256];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = a.a.C, K = a.HW * C;
    const bool bn = a.a.mode != MPNN_ACT_IDENTITY;
    if (bn) {
        for (int c = tid; c < C; c += NT_) {
            const BnC k = bn_coef(a.a, c);
            cA[c * 3] = k.m; cA[c * 3 + 1] = k.gamma * k.rstd; cA[c * 3 + 2] = k.beta;
        }
    }
    __syncthreads();
    trace_stamp(1);
    const int row = n0 + li;
    const bool valid = row < a.n;
    const int M0 = a.w[0] ? a.M[0] : 0, M1 = a.w[1] ? a.M[1] : 0;
    // The epilogue's own operands (bias, the dyn_k_cpt column) are requested NOW: loaded where they are
    // used they were one more memory round trip at the very end of the kernel.
    float ep_bias[NO], ep_extra[NO];
    bool ep_on[NO];
    int ep_row[NO], ep_col[NO], ep_M[NO], ep_s[NO];
#pragma unroll
    for (int q = 0; q < NO; ++q) {
        const int o = tid + q * NT_;             // (set, lane, r): 2 * 64 * 4 outputs
        const int e = o & 255, l = e >> 2, r = e & 3;
        ep_bias[q] = 0.f; ep_extra[q] = 0.f;
        ep_s[q] = (o >> 8) & 1;
        ep_M[q] = ep_s[q] ? M1 : M0;
        ep_row[q] = n0 + (l >> 4) * 4 + r; ep_col[q] = l & 15;
        ep_on[q] = o < 512 && ep_M[q] > 0 && ep_row[q] < a.n && ep_col[q] < ep_M[q];
        if (ep_on[q] && S == 1) {               // (sliced: only the last arriver needs them, it loads them then)
            ep_bias[q] = (ep_s[q] ? a.b[1] : a.b[0])[ep_col[q]];
            if (ep_s[q] ? a.extra_col[1] : a.extra_col[0])
                ep_extra[q] = a.alpha_cpt * a.k_cpt[ep_row[q]] * (ep_s[q] ? a.w[1] : a.w[0])[(size_t)K * ep_M[q] + ep_col[q]];
        }
    }
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    // LF_UN 16-feature blocks per iteration, ALL their loads issued before the first MFMA: a wave's K
    // share is a chain of dependent memory round trips (one per iteration), 8 of them at K = 2048 with
    // one block per iteration -- two with four.
    constexpr int LF_UN = 4;
    const int nkb_all = K >> 4;
    const int kb_lo = slice * nkb_all / S, nkb = (slice + 1) * nkb_all / S;      // this workgroup's blocks [kb_lo, nkb)
    const float *xrow = a.a.x + (size_t)(valid ? row : 0) * K;
    for (int kb = kb_lo + wid * LF_UN; kb < nkb; kb += LFW * LF_UN) {
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
    float vsum[NO];
#pragma unroll
    for (int q = 0; q < NO; ++q) {
        const int o = tid + q * NT_, e = o & 255;
        float v = 0.f;
        if (o < 512) {
#pragma unroll
            for (int w = 0; w < LFW; ++w) v += red[(w * 2 + ep_s[q]) * 256 + e];
        }
        vsum[q] = v;
    }
    if (S == 1) {
#pragma unroll
        for (int q = 0; q < NO; ++q)
            if (ep_on[q]) (ep_s[q] ? a.y[1] : a.y[0])[(size_t)ep_row[q] * ep_M[q] + ep_col[q]] = (ep_bias[q] + vsum[q]) + ep_extra[q];
    } else if constexpr (SLICED) {
        __shared__ int ticket;
        float *part = a.kpart + (size_t)blockIdx.x * MPNN_LIN_KSLICES * 512;
#pragma unroll
        for (int q = 0; q < NO; ++q) {
            const int o = tid + q * NT_;
            if (o < 512) __hip_atomic_store(part + slice * 512 + o, vsum[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // every wave waits until ITS write-through stores have been acknowledged (vmcnt(0)) before the
        // barrier in front of the ticket: the workgroup-scope fence alone lowers to lgkmcnt(0) only, and the
        // ticket could then become visible before other waves' partials (different L2 channels)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        if (tid == 0) ticket = __hip_atomic_fetch_add(a.kcnt + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __syncthreads();
        if (ticket == S - 1) {
#pragma unroll
            for (int q = 0; q < NO; ++q) {
                if (!ep_on[q]) continue;
                const int o = tid + q * NT_;
                float pv[MPNN_LIN_KSLICES];
#pragma unroll
                for (int sl = 0; sl < MPNN_LIN_KSLICES; ++sl)
                    pv[sl] = __hip_atomic_load(part + (sl < S ? sl : 0) * 512 + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                float v = (ep_s[q] ? a.b[1] : a.b[0])[ep_col[q]];
#pragma unroll
                for (int sl = 0; sl < MPNN_LIN_KSLICES; ++sl) v += sl < S ? pv[sl] : 0.f;
                if (ep_s[q] ? a.extra_col[1] : a.extra_col[0])
                    v += a.alpha_cpt * a.k_cpt[ep_row[q]] * (ep_s[q] ? a.w[1] : a.w[0])[(size_t)K * ep_M[q] + ep_col[q]];
                (ep_s[q] ? a.y[1] : a.y[0])[(size_t)ep_row[q] * ep_M[q] + ep_col[q]] = v;
            }
            if (tid == 0) __hip_atomic_store(a.kcnt + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    trace_stamp(5);
}

int mpnn_trace_install_lin(void *buf) { return mpnn_trace_install(buf); }

extern "C" int mpnn_lin_fwd(const mpnn_lin_fwd_args *dev_table, int count, int n_max, void *stream) {
    if (count <= 0 || n_max <= 0) return 0;
    if (!dev_table) return MPNN_E_ARG;
    hipLaunchKernelGGL((lin_fwd_k<16, false>), dim3((n_max + 15) / 16, count), dim3(16 * 64), 0, (hipStream_t)stream, dev_table, count);
    MPNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mpnn_lin_fwd_ks(const mpnn_lin_fwd_args *dev_table, int count, int n_max, int k_max, void *stream) {
    if (count <= 0 || n_max <= 0) return 0;
    if (!dev_table || k_max <= 0) return MPNN_E_ARG;
    int s_max = k_max >> 8;                                  // slices of the largest record: the grid's extent
    s_max = s_max < 1 ? 1 : (s_max > MPNN_LIN_KSLICES ? MPNN_LIN_KSLICES : s_max);
    hipLaunchKernelGGL((lin_fwd_k<4, true>), dim3((n_max + 15) / 16, count * s_max), dim3(4 * 64), 0,
                       (hipStream_t)stream, dev_table, count);
    MPNN_LAUNCH_CHECK();
    return 0;
}

// ------------------------------- backward -----------------------------------
// A workgroup owns 64 features k (one 16-feature MFMA tile per wave) and ALL batch rows, 16 per pass;
// both contractions run on v_mfma_f32_16x16x4_f32:
//   dX[r][k]    = sum_m dY[r][m] * W[k][m]          D[i = r][j = k]   inner index m = 4s + g, s = 0..7
//   dW[k][m]    = sum_r act(X)[r][k] * dY[r][m]      D[i = k][j = m]   inner index r = 4g + s, s = 0..3
//   db[m]       = sum_r dY[r][m]
// (m: the head's outputs in columns 0..15, the router's in 16..31, zero-padded.)  A lane (g, li) of the
// dX tile holds rows 4g..4g+3 of feature li -- exactly the A operands the dW contraction wants when its
// inner index is ordered r = 4g + s, so X is loaded ONCE, in that layout (every row of the batch up
// front: one memory round trip), W straight into B-operand registers, and only dY goes through LDS.
// No workgroup shares an output with another: dW, db are plain stores in a fixed summation order.
// (History: a thread-per-feature VALU loop, ~100 instructions per row, with the rows split over eight
// workgroups that ADDED their dW partials -- 0.8 M fp32 atomics per step, 7 us of a 21 us launch.)
//
// Row split (mpnn_lin_bwd_rs, small batches): gridDim.z workgroups share the rows of a feature block, four
// passes (64 rows) at a time each; they leave their dW / db partial tiles in scratch with write-through
// stores and take a ticket, and the last to arrive adds the partials in row-group order (as
// mpnn_lin_fwd_ks: deterministic, no agent-scope fence).  The unsplit form reads 52 KB per workgroup
// in 56 loads per thread before its first MFMA: 6 of its 17 us.
#define LB_ROWS 16          // rows per MFMA pass
#define LB_DP 36            // LDS pitch of a dY row: 4 * 36 = 16 (mod 32) -> the dW operand reads are conflict-free
// NP = passes (of 16 rows) held in LDS / registers at a time; RS = row split
template <int NP, bool RS>
__global__ __launch_bounds__(256) void lin_bwd_k(const mpnn_lin_bwd_args *__restrict__ tab) {
    constexpr int LB_SUPER = NP * LB_ROWS, LB_NP = NP, LB_GP = NP < 4 ? NP : 4;     // LB_GP: passes in flight together
    const mpnn_lin_bwd_args &a = tab[blockIdx.y];      // (by reference: a copy needs more SGPRs than there are -- 254 spills, +1.2 us)
    const int C = a.a.C, K = a.HW * C;
    const bool has_extra = a.extra_col[0] || a.extra_col[1];
    const int kext = K + (has_extra ? 1 : 0);
    const int k0 = blockIdx.x * 64;
    if (k0 >= kext) return;
    // this workgroup's rows [r_lo, r_hi): whole passes, dealt evenly to the Z row groups that have any
    const int np_all = (a.n + LB_ROWS - 1) / LB_ROWS;
    const int ppw = RS ? (np_all + (int)gridDim.z - 1) / (int)gridDim.z : np_all;
    const int Z = RS ? (np_all + ppw - 1) / max(ppw, 1) : 1;
    const int zi = RS ? (int)blockIdx.z : 0;
    if (zi >= Z) return;
    const int r_lo = zi * ppw * LB_ROWS, r_hi = min(a.n, (zi + 1) * ppw * LB_ROWS);
    trace_stamp(0); trace_note(6, 11);
    __shared__ float dys[LB_SUPER * LB_DP];
    __shared__ __attribute__((aligned(16))) float coef[64 * 4];       // per local feature: mean, gamma*rstd, beta, rstd
    __shared__ float tr[2 * 4 * 64 + 256];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M0 = a.w[0] ? a.M[0] : 0, M1 = a.w[1] ? a.M[1] : 0;
    const bool bn = a.a.mode != MPNN_ACT_IDENTITY;
    const bool fuse_bn = a.dz_out != nullptr && bn;          // uniform: fused mpnn_bn_bwd_reduce
    const int fl = wid * 16 + li, f = k0 + fl;               // this lane's feature
    // ---- one memory round trip for everything read before the arithmetic ----
    constexpr int DY_PT = LB_SUPER * 32 / 256;
    float xv[LB_NP][4], dyr[DY_PT];
    auto load_super = [&](int R0) {
        const int nr = min(LB_SUPER, r_hi - R0);
#pragma unroll
        for (int q = 0; q < DY_PT; ++q) {
            const int i = tid + q * 256, rr = i >> 5, col = i & 31, s = col >> 4, m = col & 15;
            const int M = s ? M1 : M0;
            const bool ok = rr < nr && m < M;
            const float *dyp = s ? a.dy[1] : a.dy[0];
            dyr[q] = dyp ? dyp[ok ? (size_t)(R0 + rr) * M + m : 0] : 0.f;
            dyr[q] = ok ? dyr[q] : 0.f;
        }
#pragma unroll
        for (int p = 0; p < LB_NP; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rr = LB_ROWS * p + 4 * g + q;
                const bool okx = rr < nr && f < K, oke = rr < nr && f == K && has_extra;
                float v = a.a.x[okx ? (size_t)(R0 + rr) * K + f : 0];
                if (has_extra) { const float e = a.k_cpt[oke ? R0 + rr : 0]; v = oke ? a.alpha_cpt * e : v; }
                xv[p][q] = (okx || oke) ? v : 0.f;
            }
    };
    load_super(r_lo);
    if (tid < 64) {
        f32x4 cf = {0.f, 1.f, 0.f, 0.f};
        if (bn && k0 + tid < K) { const BnC c = bn_coef(a.a, (k0 + tid) % C); cf[0] = c.m; cf[1] = c.gamma * c.rstd; cf[2] = c.beta; cf[3] = c.rstd; }
        ((f32x4 *)coef)[tid] = cf;
    }
    // W in B-operand layout: wv[s] = W_set[f][4 (s & 3) + g], set = s >> 2
    float wv[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int set = s >> 2, m = 4 * (s & 3) + g, M = set ? M1 : M0;
        const float *wp = set ? a.w[1] : a.w[0];
        const bool ok = m < M && (f < K || (f == K && (set ? a.extra_col[1] : a.extra_col[0])));
        const float v = wp ? wp[ok ? (size_t)f * M + m : 0] : 0.f;
        wv[s] = ok ? v : 0.f;
    }
    f32x4 accW[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    float r1 = 0.f, r2 = 0.f;                               // sum dz, sum dz * xhat over this lane's rows
    float dbs = 0.f;
    trace_stamp(1);
    for (int R0 = r_lo; R0 < r_hi; R0 += LB_SUPER) {
        const int nr = min(LB_SUPER, r_hi - R0);
        if (R0 != r_lo) load_super(R0);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < DY_PT; ++q) { const int i = tid + q * 256; dys[(i >> 5) * LB_DP + (i & 31)] = dyr[q]; }
        __syncthreads();
        if (blockIdx.x == 0) {                               // db: eight row groups of 32 columns, fixed order
            const int col = tid & 31, rg = tid >> 5;
            float t = 0.f;
#pragma unroll
            for (int rr = 0; rr < LB_SUPER / 8; ++rr) t += dys[(rg * (LB_SUPER / 8) + rr) * LB_DP + col];   // (rows >= nr are zero)
            dbs += t;
        }
        static_assert(LB_SUPER % 8 == 0, "db row groups");
        const f32x4 c = ((const f32x4 *)coef)[fl];
        // LB_GP passes at a time: their LDS reads, MFMA chains (independent across passes) and stores overlap
        // (one pass at a time was a serial chain of LDS wait -> 8 dependent MFMAs -> drain -> stores: 1 us each)
#pragma unroll
        for (int pg = 0; pg < LB_NP; pg += LB_GP) {
            if (LB_ROWS * pg >= nr) break;                     // (uniform; rows >= nr are zero in LDS, masked below)
            float ady[LB_GP][8], bdy[LB_GP][2][4];
#pragma unroll
            for (int pp = 0; pp < LB_GP; ++pp) {
                const int p = pg + pp;
#pragma unroll
                for (int s = 0; s < 8; ++s) ady[pp][s] = dys[(LB_ROWS * p + li) * LB_DP + 4 * s + g];              // dY[r = li][m = 4s + g]
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int s = 0; s < 4; ++s) bdy[pp][mt][s] = dys[(LB_ROWS * p + 4 * g + s) * LB_DP + 16 * mt + li];   // dY[r = 4g + s][m]
            }
            float xa[LB_GP][4], xh[LB_GP][4];
            f32x4 dx[LB_GP];
#pragma unroll
            for (int pp = 0; pp < LB_GP; ++pp) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float xc = xv[pg + pp][q] - c[0];
                    xa[pp][q] = (bn && f < K) ? fmaxf(xc * c[1] + c[2], 0.f) : xv[pg + pp][q];
                    xh[pp][q] = xc * c[3];
                }
                dx[pp] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int pp = 0; pp < LB_GP; ++pp) dx[pp] = __builtin_amdgcn_mfma_f32_16x16x4f32(ady[pp][s], wv[s], dx[pp], 0, 0, 0);
#pragma unroll
            for (int pp = 0; pp < LB_GP; ++pp)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    accW[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[pp][s], bdy[pp][0][s], accW[0], 0, 0, 0);
                    accW[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[pp][s], bdy[pp][1][s], accW[1], 0, 0, 0);
                }
            // (pin the dW accumulators here: nothing reads them until the loop ends, so the compiler sank their MFMAs
            // below the drain, to the loop latch -- where the next iteration's register copies read them behind five
            // scalar instructions: the hazard mfma_drain exists for, tools/scan_mfma_hazard.py)
            asm volatile("" : "+a"(accW[0]), "+a"(accW[1]));
            mfma_drain();
#pragma unroll
            for (int pp = 0; pp < LB_GP; ++pp)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int rr = LB_ROWS * (pg + pp) + 4 * g + q;
                    if (rr < nr && f < K) {
                        if (a.dx) a.dx[(size_t)(R0 + rr) * K + f] = dx[pp][q];
                        if (fuse_bn) {
                            const float dz = xa[pp][q] > 0.f ? dx[pp][q] : 0.f;
                            a.dz_out[(size_t)(R0 + rr) * K + f] = dz;
                            r1 += dz; r2 += dz * xh[pp][q];
                        }
                    }
                }
        }
    }
    trace_stamp(4);
    // db of this workgroup's rows (feature block 0 only): eight row-group partials, fixed order
    float dbt = 0.f;
    if (blockIdx.x == 0) {
        float *dbp = tr + 512;
        dbp[tid] = dbs;
        __syncthreads();
        if (tid < 32) {
#pragma unroll
            for (int rg = 0; rg < 8; ++rg) dbt += dbp[rg * 32 + tid];
        }
    }
    if constexpr (RS) {
        if (Z > 1) {
            // partial tiles -> scratch (write-through); the last row group to arrive adds them in row-group order
            __shared__ int ticket;
            float *part0 = a.kpart + (size_t)blockIdx.x * gridDim.z * MPNN_LIN_RS_TILE;
            float *mine = part0 + (size_t)zi * MPNN_LIN_RS_TILE;
#pragma unroll
            for (int set = 0; set < 2; ++set)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    __hip_atomic_store(mine + (wid * 16 + 4 * g + q) * 32 + set * 16 + li, accW[set][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (blockIdx.x == 0 && tid < 32) __hip_atomic_store(mine + 2048 + tid, dbt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // store acknowledgement of EVERY wave before the ticket (see lin_fwd_k)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            if (tid == 0) ticket = __hip_atomic_fetch_add(a.kcnt + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __syncthreads();
            if (ticket != Z - 1) { trace_stamp(5); goto bn_sums; }
            {
                float pv[2][4][MPNN_LIN_RSPLIT];
#pragma unroll
                for (int set = 0; set < 2; ++set)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int z = 0; z < MPNN_LIN_RSPLIT; ++z)
                            pv[set][q][z] = __hip_atomic_load(part0 + (size_t)(z < Z ? z : 0) * MPNN_LIN_RS_TILE + (wid * 16 + 4 * g + q) * 32 + set * 16 + li,
                                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                float pd[MPNN_LIN_RSPLIT];
#pragma unroll
                for (int z = 0; z < MPNN_LIN_RSPLIT; ++z)
                    pd[z] = (blockIdx.x == 0 && tid < 32) ? __hip_atomic_load(part0 + (size_t)(z < Z ? z : 0) * MPNN_LIN_RS_TILE + 2048 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.f;
#pragma unroll
                for (int set = 0; set < 2; ++set)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float v = 0.f;
#pragma unroll
                        for (int z = 0; z < MPNN_LIN_RSPLIT; ++z) v += z < Z ? pv[set][q][z] : 0.f;
                        accW[set][q] = v;
                    }
                dbt = 0.f;
#pragma unroll
                for (int z = 0; z < MPNN_LIN_RSPLIT; ++z) dbt += z < Z ? pd[z] : 0.f;
                if (tid == 0) __hip_atomic_store(a.kcnt + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    // dW tile of a set: lane holds rows 4g..4g+3 of the wave's 16 features, column li
#pragma unroll
    for (int set = 0; set < 2; ++set) {
        const float *wp = set ? a.w[1] : a.w[0];
        float *dwp = set ? a.dw[1] : a.dw[0];
        if (!wp || !dwp) continue;                      // uniform
        const int M = set ? a.M[1] : a.M[0];
        const int krows = (set ? a.extra_col[1] : a.extra_col[0]) ? K + 1 : K;
        if (li < M) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int fr = k0 + wid * 16 + 4 * g + q;
                if (fr < krows) dwp[(size_t)fr * M + li] = accW[set][q];
            }
        }
    }
    if (blockIdx.x == 0 && tid < 32) {
        const int s = tid >> 4, m = tid & 15;
        const float *wps = s ? a.w[1] : a.w[0];
        float *dbp2 = s ? a.db[1] : a.db[0];
        if (wps && dbp2 && m < (s ? a.M[1] : a.M[0])) dbp2[m] = dbt;
    }
bn_sums:
    // fused BatchNorm-backward reductions: feature k = pixel * C + c; the workgroup's 64 features are
    // 64 / C pixels of all C channels (C | 64) or a 64-channel slice of one pixel
    if (fuse_bn) {
        tr[g * 64 + fl] = r1; tr[256 + g * 64 + fl] = r2;
        __syncthreads();
        float f1 = 0.f, f2 = 0.f;
        if (tid < 64) {
            f1 = (tr[tid] + tr[64 + tid]) + (tr[128 + tid] + tr[192 + tid]);
            f2 = (tr[256 + tid] + tr[320 + tid]) + (tr[384 + tid] + tr[448 + tid]);
        }
        __syncthreads();
        if (tid < 64) { tr[tid] = f1; tr[64 + tid] = f2; }
        __syncthreads();
        const int span = C < 64 ? C : 64;                    // distinct channels in this workgroup
        if (tid < span && k0 + tid < K) {
            double a1 = 0.0, a2 = 0.0;
            for (int t = tid; t < 64 && k0 + t < K; t += span) { a1 += (double)tr[t]; a2 += (double)tr[64 + t]; }
            const int c = (k0 + tid) % C;
            double *slot = a.red_out + (size_t)((blockIdx.x + zi * gridDim.x) % a.red_nslot) * 2 * C;
            atomicAdd(slot + c, a1);
            atomicAdd(slot + C + c, a2);
        }
    }
    trace_stamp(5);
}

extern "C" int mpnn_lin_bwd(const mpnn_lin_bwd_args *dev_table, int count, int n_max, int k_max, void *stream) {
    if (count <= 0 || n_max <= 0) return 0;
    if (!dev_table) return MPNN_E_ARG;
    hipLaunchKernelGGL((lin_bwd_k<8, false>), dim3((k_max + 1 + 63) / 64, count), dim3(256), 0, (hipStream_t)stream, dev_table);
    MPNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mpnn_lin_bwd_rs(const mpnn_lin_bwd_args *dev_table, int count, int n_max, int k_max, void *stream) {
    if (count <= 0 || n_max <= 0) return 0;
    if (!dev_table) return MPNN_E_ARG;
    int z = (n_max + 63) / 64;                               // four 16-row passes per workgroup
    if (z > MPNN_LIN_RSPLIT) z = MPNN_LIN_RSPLIT;
    hipLaunchKernelGGL((lin_bwd_k<4, true>), dim3((k_max + 1 + 63) / 64, count, z), dim3(256), 0, (hipStream_t)stream, dev_table);
    MPNN_LAUNCH_CHECK();
    return 0;
}
