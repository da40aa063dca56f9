// Device side of mpnn_msconv_bwd_level (bwd_level.hip; bwd_level_small.hip instantiates the variants with block 0's
// swapped-role image weight gradients in a translation unit of its own: the file took over a minute to compile).
#pragma once
#include "bwd_bodies.h"

struct BwdRec { BwdScaleP q; int gk, wide, r0, r1; };
struct BwdLevelQ { int n; int w0[MPNN_BWD_LEVEL_MAX]; int reps, wpr; };    // reps > 1: `reps` copies of the level's wpr workgroups, copy r on records tab[r * n ..]

template <int GK, int OT, bool SMALLC>
__device__ __forceinline__ void level_wgrad(const BwdRec *__restrict__ r, int l, char *smem) {
    constexpr int GS = OT * 16 + 4;
    const WgP w = r->q.w;
    const int gxw = r->q.gxw, nchw = r->q.nchw;
    const int rr = l / gxw, bx = l - rr * gxw;
    const int chunk = rr % nchw, bz = rr / nchw;
    f32x4 *tile = (f32x4 *)smem;
    float *gt = (float *)(smem + 4 * WGeom<GK>::PS * 16);
    float *cA = gt + 64 * GS;
    if (chunk >= ((w.c.a.C + 15) >> 4)) wgrad_body<GK, OT, 1>(w, tile, gt, cA, bx, chunk, bz, gxw);
    else if (SMALLC && OT == 1 && w.c.a.C <= 3) { if constexpr (SMALLC && OT == 1) wgrad_body<GK, 1, 0, true>(w, tile, gt, cA, bx, chunk, bz, gxw); }
    else                                 wgrad_body<GK, OT, 0>(w, tile, gt, cA, bx, chunk, bz, gxw);
}

template <int GK, int OTMASK, bool SMALLC>
__device__ __forceinline__ void level_member(const BwdRec *__restrict__ r, int id, char *smem) {
    const int gxh = r->q.gxh, gxv = r->q.gxv;
    const int wh = r->q.gyh * gxh, wv = r->q.gyv * gxv;
    if (id < wh) {
        const ConvP p = r->q.h;
        const int by = id / gxh, bx = id - by * gxh;
        conv_body<GK, 1, 1, 4, 1, false, EPI_DGH_BN, 1>(p, bx, by, gxh, smem);
    } else if (id < wh + wv) {
        const ConvP p = r->q.v;
        const int l = id - wh, by = l / gxv, bx = l - by * gxv;
        conv_body<GK, 1, 1, 4, 1, false, EPI_DGV, 1>(p, bx, by, gxv, smem);
    } else {
        const int l = id - wh - wv;
        if constexpr (OTMASK == 3) {
            if (r->wide) level_wgrad<GK, 4, false>(r, l, smem);
            else         level_wgrad<GK, 1, SMALLC>(r, l, smem);
        } else if constexpr (OTMASK == 2) level_wgrad<GK, 4, false>(r, l, smem);
        else level_wgrad<GK, 1, SMALLC>(r, l, smem);
    }
}

template <int GK, int OTMASK> struct LevelSmem {
    static constexpr int GS = (OTMASK & 2 ? 4 : 1) * 16 + 4;
    static constexpr int CB = ConvSmem<GK, 4, 16, 1>::BYTES;
    static constexpr int WB = 4 * WGeom<GK>::PS * 16 + 64 * GS * 4 + (128 * 3 + (OTMASK & 2 ? 4 : 1) * 16 * 5) * 4;
    static constexpr int BYTES = CB > WB ? CB : WB;
};
template <int GKMASK, int OTMASK> struct LevelSmemAll {
    static constexpr int A = (GKMASK & 1) ? LevelSmem<0, OTMASK>::BYTES : 0;
    static constexpr int B = (GKMASK & 2) ? LevelSmem<1, OTMASK>::BYTES : 0;
    static constexpr int C = (GKMASK & 4) ? LevelSmem<2, OTMASK>::BYTES : 0;
    static constexpr int BYTES = A > B ? (A > C ? A : C) : (B > C ? B : C);
};

#ifndef MPNN_OCC_LEVEL
#define MPNN_OCC_LEVEL 3     // waves per SIMD of the levels without 64-channel weight-gradient groups
#endif
// SMALLC: some member's weight gradients have a 1- or 3-channel image as operand A (block 0): those members' image chunk
// runs the swapped-role body (bwd_bodies.h); levels without such a member keep the instantiation they had.
template <int GKMASK, int OTMASK, bool SMALLC = false>
__global__ __launch_bounds__(256, (OTMASK & 2) ? 2 : MPNN_OCC_LEVEL) void bwd_level_k(const BwdRec *__restrict__ tab, const BwdLevelQ lq) {
    __shared__ __attribute__((aligned(16))) char smem[LevelSmemAll<GKMASK, OTMASK>::BYTES];
    int id = blockIdx.x;
    if (lq.reps > 1) { const int rep = id / lq.wpr; id -= rep * lq.wpr; tab += rep * lq.n; }     // (uniform)
    int m = 0, w0 = 0;
#pragma unroll
    for (int k = 1; k < MPNN_BWD_LEVEL_MAX; ++k)
        if (k < lq.n && id >= lq.w0[k]) { m = k; w0 = lq.w0[k]; }
    const BwdRec *__restrict__ r = tab + m;
    const int gk = r->gk;
    trace_note(11, m + 1);
    if constexpr ((GKMASK & 1) != 0) { if (gk == 0) { level_member<0, OTMASK, SMALLC>(r, id - w0, smem); return; } }
    if constexpr ((GKMASK & 2) != 0) { if (gk == 1) { level_member<1, OTMASK, SMALLC>(r, id - w0, smem); return; } }
    if constexpr ((GKMASK & 4) != 0) { if (gk == 2) { level_member<2, OTMASK, SMALLC>(r, id - w0, smem); return; } }
}


typedef void (*LevelKern)(const BwdRec *, const BwdLevelQ);
