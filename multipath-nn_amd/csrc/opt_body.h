// TALR + L2 + momentum update (net_types.py:24-37, tf.train.MomentumOptimizer) of one work item, shared by
// mpnn_talr_momentum_step (misc.hip) and the fused end of the backward pass mpnn_backward_finish_opt (wgrad.hip).
#pragma once
#include "common.h"

struct OptP {
    float *params, *accum;  const float *grads;  const float *node_stat, *hyp;
    int talr;  float inv_n, grad_scale;  const float *w_eq;  float *packs;
};

// learning-rate scale and mean p_tr of a tree node (net_types.py:25-34)
__device__ __forceinline__ void opt_node(const OptP &o, int node, int is_router, float &scale, float &pbar) {
    pbar = o.node_stat[node * 2] * o.inv_n;                       // mean p_tr over the batch
    scale = 1.f;
    if (o.talr) scale = 1.f / sqrtf(o.node_stat[node * 2 + 1] * o.inv_n);   // 1/sqrt(mean p_tr^2)
    // a router's parameters: alpha_rtr * lr_scale, and lr_scale is 1 without TALR (net_types.py:25-33) -- the factor
    // does NOT depend on talr.  SRNets have no routers (is_router = 0 for every item).
    if (is_router) scale *= o.hyp[MPNN_HYP_ARTR];
}

// one element without a weight pack (BatchNorm gamma / beta in the fused launch): the same arithmetic as opt_seg
__device__ __forceinline__ void opt_elem(const OptP &o, int off, float graw, float l2, float scale, float pbar) {
    const float lr = o.hyp[MPNN_HYP_LR], mu = o.hyp[MPNN_HYP_MU];
    const float w = o.params[off];
    float g = graw * o.grad_scale;
    if (l2 != 0.f) g += 2.f * l2 * pbar * w;
    g *= scale;
    const float a = mu * o.accum[off] + g;
    o.accum[off] = a;
    o.params[off] = w - lr * a;
}

// s: MPNN_SEG_INTS ints (include/mpnn_hip.h): offset, count (<= 2048), node, is_router, l2 as float bits, w_eq offset | -1,
// and for conv weights the tensor's base, Cin, Cout and the offsets of its forward / backward packs.
// gl != nullptr: gradient element i of the item is gl[i] (LDS: the workgroup has just reduced it from the slabs)
// instead of o.grads[offset + i].  wl: 2048 floats of LDS (pack staging).
__device__ __forceinline__ void opt_seg(const OptP &o, const int *__restrict__ s, const float *gl, float *wl) {
    float *__restrict__ params = o.params, *__restrict__ accum = o.accum, *__restrict__ packs = o.packs;
    const float *__restrict__ grads = o.grads, *__restrict__ w_eq = o.w_eq, *__restrict__ hyp = o.hyp;
    const float grad_scale = o.grad_scale;
    const int off = s[0], cnt = s[1], node = s[2], is_router = s[3];
    const float *eq = (w_eq && s[5] >= 0) ? w_eq + s[5] : nullptr;      // identity part of a `res` layer
    // conv weights: the updated value also goes to its slots of the k-interleaved forward pack and of the
    // transposed, tap-flipped backward pack (mpnn_pack_weights' layout), so the next step starts with
    // current packs and no packing launch
    const int tbase = s[6], Cin = s[7], Cout = s[8], fwd = s[9], bwd = s[10];
    const bool emit = packs && Cin > 0;
    const int per_f = ((Cin + 15) >> 4) * 16 * Cout, per_b = ((Cout + 15) >> 4) * 16 * Cin, cc = Cin * Cout;
    // Fast path: the segment is a whole number of 4-row groups of [tap * Cin + ci][Cout] inside one tap or a
    // whole number of taps (every shipped shape with Cin % 4 == 0).  The updated weights meet in LDS and
    // leave as CONTIGUOUS runs of both packs (element by element the pack stores are 4-byte scatters at a
    // 16-byte stride: the kernel took twice as long as the plain update).
    const int R = emit ? cnt / Cout : 0, row0 = emit ? (off - tbase) / Cout : 0;
    const bool fast = emit && cnt <= 2048 && (Cin & 3) == 0 && (row0 & 3) == 0 && (R & 3) == 0 && R * Cout == cnt &&
                      ((R <= Cin && (row0 % Cin) + R <= Cin) || (R % Cin == 0 && row0 % Cin == 0));
    const float l2 = __int_as_float(s[4]);
    const float lr = hyp[MPNN_HYP_LR], mu = hyp[MPNN_HYP_MU];
    float scale, pbar;
    opt_node(o, node, is_router, scale, pbar);
    // A segment has at most TM_Q * 256 elements (host: 2048).  Every operand of the thread's TM_Q elements is
    // requested before the first is used (clamped addresses, no control flow): one memory round trip per
    // workgroup -- as a rolled loop over `cnt` each iteration's three loads waited for the previous
    // iteration's stores (12 us for 8 MB; the loop stays for longer segments).
    constexpr int TM_Q = 8;
    for (int i0 = 0; i0 < cnt; i0 += TM_Q * 256) {
        float wq[TM_Q], gq[TM_Q], aq[TM_Q], eqv[TM_Q];
#pragma unroll
        for (int q = 0; q < TM_Q; ++q) {
            const int i = i0 + threadIdx.x + q * 256, ic = i < cnt ? i : 0;
            wq[q] = params[off + ic]; gq[q] = gl ? gl[ic] : grads[off + ic]; aq[q] = accum[off + ic];
            eqv[q] = eq ? eq[ic] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < TM_Q; ++q) {
            const int i = i0 + threadIdx.x + q * 256;
            if (i >= cnt) continue;
            const float w = wq[q];
            float g = gq[q] * grad_scale;
            if (l2 != 0.f) g += 2.f * l2 * pbar * (w - eqv[q]);
            g *= scale;
            const float a = mu * aq[q] + g;
            accum[off + i] = a;
            const float wn = w - lr * a;
            params[off + i] = wn;
            if (fast) wl[i] = wn;
            else if (emit) {
                const int e = off + i - tbase, tap = e / cc, rem = e - tap * cc, ci = rem / Cout, co = rem - ci * Cout;
                if (fwd >= 0) packs[fwd + tap * per_f + ((ci >> 2) * Cout + co) * 4 + (ci & 3)] = wn;
                if (bwd >= 0) packs[bwd + (8 - tap) * per_b + ((co >> 2) * Cin + ci) * 4 + (co & 3)] = wn;
            }
        }
    }
    if (fast) {                                        // (uniform)
        __syncthreads();
        if (fwd >= 0) {
            const int gsz = 4 * Cout;                  // a group of 4 input channels x Cout: one contiguous pack block
            for (int p = threadIdx.x; p < cnt; p += 256) {
                const int grp = p / gsz, within = p - grp * gsz, co = within >> 2, j = within & 3;
                const int row = row0 + grp * 4, tap = row / Cin, ci0 = row - tap * Cin;
                packs[fwd + tap * per_f + (ci0 >> 2) * gsz + within] = wl[(grp * 4 + j) * Cout + co];
            }
        }
        if (bwd >= 0) {
            const int Rt = R < Cin ? R : Cin, tsz = Rt * Cout, bsz = Rt * 4;     // rows of one tap in this segment
            for (int p = threadIdx.x; p < cnt; p += 256) {
                const int tl = p / tsz, rem = p - tl * tsz, gq = rem / bsz, rem2 = rem - gq * bsz, cil = rem2 >> 2, j = rem2 & 3;
                const int row = row0 + tl * Rt + cil, tap = row / Cin, ci = row - tap * Cin;
                packs[bwd + (8 - tap) * per_b + (gq * Cin + ci) * 4 + j] = wl[(tl * Rt + cil) * Cout + 4 * gq + j];
            }
        }
    }
}

