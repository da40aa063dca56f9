// Device bodies shared by the backward launches (wgrad.hip: one scale per launch; bwd_level.hip: one
// dependency level per launch): the weight-gradient body and the argument records.
#pragma once
#include <type_traits>
#ifndef MPNN_WG_SETS
#define MPNN_WG_SETS 1
#endif
#ifndef MPNN_WG_NINE
#define MPNN_WG_NINE 1       // 0: the tap-slot form of the weight-gradient MFMA loop (A/B builds)
#endif
#ifndef MPNN_WG_SMALLC
#define MPNN_WG_SMALLC 1     // 0: block 0's image chunk in the general nine-tap form (A/B builds)
#endif
#include "conv_kernel.h"

struct WgP {
    ConvP c;                 // a, v, Cv, n, H, W, Cout
    const float *g;
    const float *g_s;  mpnn_act g_bn;  const double *g_red;  int g_nslot;  int g_on;   // g = bn_bwd_apply(dz) on load
    float *dwa, *dwv, *db;   // partial-sum destinations of split 0
    long split_stride;       // floats between consecutive splits' destinations
    int n_tiles;
};

template <int GK> struct WGeom;
template <> struct WGeom<0> { static constexpr int PS = 113; };
template <> struct WGeom<1> { static constexpr int PS = 113; };
template <> struct WGeom<2> { static constexpr int PS = 145; };

// SMALLC (PART == 0, OT == 1, a 1- or 3-channel image as operand A: block 0): the contraction with the roles swapped --
// D[i = cout][j = (tap, c)] += sum_pixels g[pixel][cout] * x[pixel + tap][c], 9 C <= 27 columns = two N-tiles -- EIGHT
// MFMAs per wave and tile instead of 36: in the general form 13 of the 16 rows of every tap's M-tile are padding, and the
// launch that is nothing but this body (h32 3 -> 16) was bound by the MFMA pipe at 14 TFLOP/s of useful work.
template <int GK, int OT, int PART, bool SMALLC = false>
__device__ __forceinline__ void wgrad_body(const WgP &p, f32x4 *tile, float *gt, float *cA,
                                           const int bx, const int by, const int bz, const int gx) {
    static_assert(!SMALLC || (PART == 0 && OT == 1 && MPNN_WG_NINE), "SMALLC: the image chunk of a 16-channel group");
    using G = Geom<GK>;
    constexpr int PS = WGeom<GK>::PS, R = G::R, HR = G::TH + 2;
    constexpr int GS = OT * 16 + 4;                  // g tile row stride (floats)
    constexpr int XN = XItems<GK>::N;
    const ConvP &c = p.c;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);        // provably wave-uniform
    const int g = lane >> 4, li = lane & 15;
    const int nchA = (c.a.C + 15) >> 4;
    constexpr int part = PART;
    const int ch = part ? by - nchA : by;
    const int C = part ? c.Cv : c.a.C;
    const int co0 = bz * OT * 16;
    int np = (C - ch * 16 + 3) >> 2;
    np = np > 4 ? 4 : np;
    const bool bias_wave = wid == 1 && by == 0;           // slot ti = 2 of wave 1 is tap 9: unused
    // tile sequence (see conv_body): XCD-aware when the split is a multiple of 8
    const bool xa = c.xcd != 0 && (c.n & 31) == 0 && (gx & 7) == 0;      // (uniform)
    const int xcd_id = blockIdx.x & 7;
    const int tpi = GK == 0 ? (c.W >> 4) * (c.H >> 2) : 1;
    const int sq0 = xa ? (bx >> 3) : bx, sqd = xa ? (gx >> 3) : gx, sqn = xa ? (p.n_tiles >> 3) : p.n_tiles;
    trace_stamp(0);

    // NINE: every wave accumulates ALL nine taps -- of its own 16-channel output tile (OT == 4: wave w owns
    // output tile w) or of its own quarter of the tile's pixels (OT == 1: wave w owns the 16-pixel group w; the
    // four partial sums meet in LDS at the end).  36 * OT MFMAs per wave and tile, all of them useful; the
    // tap-slot form (waves own taps {w, w+4, w+8}: twelve slots for nine taps) issued 48 * OT.  The bias
    // gradient is a plain sum of the g values the wave reads anyway.
    constexpr bool NINE = (OT == 1 || OT == 4) && MPNN_WG_NINE;
    f32x4 acc[NINE ? 1 : 3][OT];
    f32x4 acc9[(NINE && !SMALLC) ? 9 : 1];
    [[maybe_unused]] f32x4 accS[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    // SMALLC: column idx = nt * 16 + li of the two N-tiles = (tap, c) pair idx = tap * C + c; its LDS offset from the
    // pixel's slot (floats) and whether it exists
    [[maybe_unused]] int offS[2];
    [[maybe_unused]] bool onS[2];
    if constexpr (SMALLC) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int idx = nt * 16 + li, n9 = 9 * c.a.C;
            onS[nt] = idx < n9;
            const int ic = onS[nt] ? idx : 0, tap = ic / c.a.C, cc = ic - tap * c.a.C;
            offS[nt] = ((tap / 3) * R + (tap % 3)) * 4 + cc;          // (plane 0: channels 0..3 of the chunk)
        }
    }
    float bsum = 0.f;
#pragma unroll
    for (int ti = 0; ti < (NINE ? 1 : 3); ++ti)
#pragma unroll
        for (int nt = 0; nt < OT; ++nt) acc[ti][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < ((NINE && !SMALLC) ? 9 : 1); ++tap) acc9[tap] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *tf = (const float *)tile;
    const int a_lane = (li >> 2) * PS * 4 + (li & 3);      // plane + component of channel li
    const float one_hot = li == 0 ? 1.f : 0.f;
    // Tap slots of this wave: {wid, wid+4, wid+8}; slot 2 of waves 1-3 has no tap (9..11): it runs a
    // clamped duplicate whose result is dropped (wave 1 / chunk 0 uses it for the bias gradient), so
    // the hot loop is branch-free and every wave issues the same 3 MFMAs per step.
    int tap_off[3];
#pragma unroll
    for (int ti = 0; ti < 3; ++ti) {
        const int tap = min(wid + 4 * ti, 8);
        tap_off[ti] = (tap / 3) * R + (tap % 3);
    }

    [[maybe_unused]] int slot9[4];                        // NINE, OT == 1: LDS slots of pixels 4g + j of pixel group `wid`
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int img, ty, tx;
        mtile_pix<GK>(wid, 4 * g + j, img, ty, tx);
        slot9[j] = (img * HR + ty) * R + tx;
    }

    if (part == 0 && c.a.mode != MPNN_ACT_IDENTITY) {
        for (int cc = tid; cc < c.a.C; cc += 256) {
            const BnC k = bn_coef(c.a, cc);
            cA[cc * 3] = k.m; cA[cc * 3 + 1] = k.gamma * k.rstd; cA[cc * 3 + 2] = k.beta;
        }
    }

    float *cG = cA + 128 * 3;                        // [OT*16][5]: BatchNorm-backward coefficients of this cout group
    if (p.g_on && p.g_bn.mode == MPNN_ACT_BN_BATCH) {    // (training: every input of a row in one round trip)
        for (int cc = tid - 128; cc >= 0 && cc < OT * 16; cc += 256) bn_bwd_row(p.g_bn, p.g_red, p.g_nslot, co0 + cc, false, cG + cc * 5);
    } else if (p.g_on) {
        const double inv = 1.0 / (double)p.g_bn.cnt;
        for (int cc = tid - 128; cc >= 0 && cc < OT * 16; cc += 256) {     // waves 2-3: beside the table above
            const int co = co0 + cc;
            const BnC k = bn_coef(p.g_bn, co);
            float *e = cG + cc * 5;
            e[0] = k.m; e[1] = k.rstd; e[2] = k.gamma * k.rstd;
            double r0, r1;
            slot_sum2(p.g_red, 2 * c.Cout, co, c.Cout + co, p.g_nslot, r0, r1);
            e[3] = (float)(r0 * inv); e[4] = (float)(r1 * inv);
        }
    }

    if constexpr (XItems<GK>::INTERIOR) zero_halo(tile, 4 * PS, tid, 256);       // the halo ring: zero for the whole kernel
    __syncthreads();
    trace_stamp(1);
    // Two register sets, prefetch distance two tiles: while tile t is in LDS under the MFMAs, tile
    // t + gx is landed / landing in the other set and tile t + 2 gx is requested into the set that was
    // just written to LDS.  (With one set the loop waited a full memory round trip per tile.)
    // (64-channel groups, OT > 1, keep one set: two would not fit the register file.)
    constexpr int NS = OT == 1 ? MPNN_WG_SETS : 1;   // register sets = prefetch distance in tiles
    f32x4 xrS[NS][XN][1], grS[NS][OT], gsS[NS][OT];
    int on0[NS];
    unsigned inbS[NS];                                // in-bounds bits of the x items of the tile held in set S
    // Lean staging (as conv_body): what does not depend on the tile -- an item's LDS slot and halo pixel, a g item's
    // pixel and channel quad -- is computed ONCE per kernel; a tile costs a few multiply-adds per item.
    ItemK<GK> ik;
    item_consts<GK, PS>(ik, tid);
    int g_geo[OT], g_lds[OT], g_c4[OT];               // img << 16 | ty << 8 | tx ; float index in gt ; channel offset
#pragma unroll
    for (int k = 0; k < OT; ++k) {
        const int i = tid + k * 256;                   // 64 * OT * 4 items
        const int q = i % (OT * 4), pi = i / (OT * 4);
        int img, ty, tx;
        mtile_pix<GK>(pi >> 4, pi & 15, img, ty, tx);
        g_geo[k] = (img << 16) | (ty << 8) | tx;
        g_lds[k] = pi * GS + q * 4;
        g_c4[k] = q * 4;
    }
    const int xc = ch * 16 + ik.q * 4;                 // first channel of this thread's x items
    const bool xq_in = ik.q < np;
    const int sh = PART == 0 ? c.a.shift : 0;          // ToPyramid's strided pick (block 0); 0 elsewhere
    const int xC = PART == 0 ? c.a.C : c.Cv;
    const float *const xsrc = PART == 0 ? c.a.x : c.v;
    auto request = [&](auto sel, int t) {
        constexpr int S = decltype(sel)::value;
        int n0, y0, x0;
        tile_origin<GK>(c, xa ? xcd_tile<GK>(t, xcd_id, tpi) : t, n0, y0, x0);
        on0[S] = n0;
        unsigned inb = 0;
#pragma unroll
        for (int k = 0; k < XN; ++k) {
            const int n = n0 + (ik.geo[k] >> 16), y = y0 + ((ik.geo[k] >> 8) & 255) - 1, x = x0 + (ik.geo[k] & 255) - 1;
            const bool ok = ((ik.ok >> k) & 1) && xq_in && n < c.n &&
                            (XItems<GK>::INTERIOR || ((unsigned)y < (unsigned)c.H && (unsigned)x < (unsigned)c.W));
            inb |= (ok ? 1u : 0u) << k;
            // unconditional loads from a clamped address (no branch -> no vmcnt wait between items)
            const unsigned pix = ok ? (((unsigned)n * (c.H << sh) + (y << sh)) * (c.W << sh) + (x << sh)) * xC : 0u;
            if (PART == 1 || (xC & 3) == 0) {          // (uniform)
                xrS[S][k][0] = *(const f32x4 *)((const char *)xsrc + (pix + (ok ? xc : 0)) * 4u);
            } else {                                   // raw image with 1 or 3 channels: clamped scalar loads
#pragma unroll
                for (int j = 0; j < 4; ++j) xrS[S][k][0][j] = xsrc[pix + (ok && xc + j < xC ? xc + j : 0)];
            }
        }
        inbS[S] = inb;
#pragma unroll
        for (int k = 0; k < OT; ++k) {
            const int n = n0 + (g_geo[k] >> 16);
            const bool live = n < c.n;
            const unsigned off = live ? (((unsigned)n * c.H + y0 + ((g_geo[k] >> 8) & 255)) * c.W + x0 + (g_geo[k] & 255)) * c.Cout + co0 + g_c4[k] : 0u;
            grS[S][k] = *(const f32x4 *)((const char *)p.g + off * 4u);          // raw: out-of-range images are zeroed when stored
            if (p.g_on) gsS[S][k] = *(const f32x4 *)((const char *)p.g_s + off * 4u);    // (uniform)
        }
    };
    auto tile_step = [&](auto sel, int t) {
        constexpr int S = decltype(sel)::value;
        f32x4 (*xr)[1] = xrS[S];
        f32x4 *gr = grS[S], *gs = gsS[S];
        const int o_n0 = on0[S];
        const unsigned inb = inbS[S];
        lds_barrier();                                 // previous tile's LDS reads are done
#pragma unroll
        for (int k = 0; k < XN; ++k) {
            const bool live = (inb >> k) & 1;
            f32x4 v = xr[k][0];
            if (PART == 0) {
                if (c.a.mode != MPNN_ACT_IDENTITY) {   // (uniform)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float *cc = cA + (xc + j) * 3;
                        const float tv = fmaxf((v[j] - cc[0]) * cc[1] + cc[2], 0.f);
                        v[j] = (xc + j < xC) ? tv : 0.f;
                    }
                } else if (xC & 3) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (xc + j < xC) ? v[j] : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = live ? v[j] : 0.f;
            if ((ik.ok >> k) & 1) tile[ik.slot[k]] = v;
        }
#pragma unroll
        for (int k = 0; k < OT; ++k) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (o_n0 + (g_geo[k] >> 16) < c.n) {       // (out-of-range images stay exactly zero)
                v = gr[k];
                if (p.g_on) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float *e = cG + (g_c4[k] + j) * 5;
                        const float xh = (gs[k][j] - e[0]) * e[1];
                        v[j] = e[2] * (v[j] - e[3] - xh * e[4]);
                    }
                }
            }
            *(f32x4 *)(gt + g_lds[k]) = v;
        }
        lds_barrier();
        if (t == sq0) trace_stamp(2);
        if (t + NS * sqd < sqn) request(sel, t + NS * sqd);          // flies under NS tiles of MFMAs
        mfma_prio_on();
        if constexpr (SMALLC) {
            float gq[4], xq[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gq[j] = gt[(wid * 16 + 4 * g + j) * GS + li];          // A[i = cout li][k = g]: pixel 4 g + j of this wave's group
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) xq[j][nt] = tf[slot9[j] * 4 + offS[nt]];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bsum += gq[j];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    accS[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(gq[j], onS[nt] ? xq[j][nt] : 0.f, accS[nt], 0, 0, 0);
            }
        } else
        if constexpr (NINE) {
#pragma unroll
            for (int kk = 0; kk < (OT == 4 ? 4 : 1); ++kk) {
                const int kc = OT == 4 ? kk : wid;             // OT == 1: this wave's pixel group
                float bq[4], aq[4][9];
                if constexpr (OT == 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int img, ty, tx;
                        mtile_pix<GK>(kk, 4 * g + j, img, ty, tx);
                        const int slot = (img * HR + ty) * R + tx;
                        bq[j] = gt[(kk * 16 + 4 * g + j) * GS + wid * 16 + li];
#pragma unroll
                        for (int tap = 0; tap < 9; ++tap) aq[j][tap] = tf[(slot + (tap / 3) * R + (tap % 3)) * 4 + a_lane];
                    }
                } else {
                    // (the pixel group is a run-time value: the slot of pixel 4g + j of group `wid` comes from a small table)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bq[j] = gt[(kc * 16 + 4 * g + j) * GS + li];
#pragma unroll
                        for (int tap = 0; tap < 9; ++tap) aq[j][tap] = tf[(slot9[j] + (tap / 3) * R + (tap % 3)) * 4 + a_lane];
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bsum += bq[j];
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap)
                        acc9[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[j][tap], bq[j], acc9[tap], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            float bq[4][OT], aq[4][3];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int img, ty, tx;
                mtile_pix<GK>(kc, 4 * g + j, img, ty, tx);
                const int slot = (img * HR + ty) * R + tx;
#pragma unroll
                for (int nt = 0; nt < OT; ++nt) bq[j][nt] = gt[(kc * 16 + 4 * g + j) * GS + nt * 16 + li];
#pragma unroll
                for (int ti = 0; ti < 3; ++ti) aq[j][ti] = tf[(slot + tap_off[ti]) * 4 + a_lane];
                if (bias_wave) aq[j][2] = one_hot;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ti = 0; ti < 3; ++ti)
#pragma unroll
                    for (int nt = 0; nt < OT; ++nt)
                        acc[ti][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[j][ti], bq[j][nt], acc[ti][nt], 0, 0, 0);
        }
        }
        mfma_prio_off();
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, NS - 1>;
    if (sq0 < sqn) request(S0{}, sq0);
    if (NS == 2 && sq0 + sqd < sqn) request(S1{}, sq0 + sqd);
    for (int t = sq0; t < sqn; t += NS * sqd) {
        tile_step(S0{}, t);
        if (NS == 2 && t + sqd < sqn) tile_step(S1{}, t + sqd);
    }

    trace_stamp(4);
    mfma_drain();
    // D layout: col = li (cout), row = g*4 + r (input channel of the chunk).
    const size_t soff = (size_t)bx * p.split_stride;
    float *dw = (part ? p.dwv : p.dwa) + soff;
    if constexpr (NINE && OT == 4) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cin = ch * 16 + g * 4 + r;
                if (cin < C) dw[((size_t)tap * C + cin) * c.Cout + co0 + wid * 16 + li] = acc9[tap][r];
            }
        if (by == 0) {                                  // db = sum over the pixels of g: lanes li, li+16, li+32, li+48 hold the four pixel quarters
            float b = bsum;
            b += __shfl_xor(b, 16);
            b += __shfl_xor(b, 32);
            if (g == 0) p.db[soff + co0 + wid * 16 + li] = b;
        }
    } else if constexpr (SMALLC) {
        // D rows = cout (4 g + r), columns = (tap, c) pairs; the four waves' partial sums (their pixel groups) meet in LDS,
        // summed in wave order; dw[(tap * C + c) * Cout + cout] = dw[idx * Cout + cout]
        f32x4 *part4 = tile;                             // [wave][N-tile][lane]
        float *partf = (float *)tile;
        lds_barrier();                                   // the MFMA reads of the last tile are done
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) part4[(wid * 2 + nt) * 64 + lane] = accS[nt];
        lds_barrier();
        {
            const int r = wid, n9 = 9 * C;               // thread = (component r, lane): two N-tiles each
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) v += partf[((w * 2 + nt) * 64 + lane) * 4 + r];
                const int idx = nt * 16 + li, co = 4 * g + r;
                if (idx < n9) dw[(size_t)idx * c.Cout + co0 + co] = v;
            }
        }
        if (by == 0) {
            float b = bsum;
            b += __shfl_xor(b, 16);
            b += __shfl_xor(b, 32);
            lds_barrier();
            if (g == 0) partf[wid * 16 + li] = b;
            lds_barrier();
            if (tid < 16) p.db[soff + co0 + tid] = (partf[tid] + partf[16 + tid]) + (partf[32 + tid] + partf[48 + tid]);
        }
    } else if constexpr (NINE) {
        // OT == 1: the four waves hold partial sums over their pixel groups; they meet in LDS three taps at a time
        // (the tile and g buffers are free now: 3 taps x 4 waves x 1 KB), summed in wave order.
        f32x4 *part4 = tile;                             // [wave][tap of the round][lane]
        float *partf = (float *)tile;
        const int cl = (tid >> 4) & 15, tl = tid >> 8;  // (256 threads: tl == 0; the round loop covers the three taps)
        (void)tl;
#pragma unroll
        for (int rd = 0; rd < 3; ++rd) {
            lds_barrier();                               // the MFMA reads of the last tile / the previous round's sums are done
#pragma unroll
            for (int q = 0; q < 3; ++q) part4[(wid * 3 + q) * 64 + lane] = acc9[rd * 3 + q];
            lds_barrier();
            const int cin = ch * 16 + cl;                // cl = 4 g + r: lane g * 16 + li, component r of the accumulator
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) v += partf[((w * 3 + q) * 64 + (cl >> 2) * 16 + li) * 4 + (cl & 3)];
                if (cin < C) dw[((size_t)(rd * 3 + q) * C + cin) * c.Cout + co0 + li] = v;
            }
        }
        if (by == 0) {
            float b = bsum;
            b += __shfl_xor(b, 16);
            b += __shfl_xor(b, 32);
            lds_barrier();
            if (g == 0) partf[wid * 16 + li] = b;
            lds_barrier();
            if (tid < 16) p.db[soff + co0 + tid] = (partf[tid] + partf[16 + tid]) + (partf[32 + tid] + partf[48 + tid]);
        }
    } else {
#pragma unroll
    for (int ti = 0; ti < 3; ++ti) {
        const int tap = wid + 4 * ti;
        if (tap >= 9) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cin = ch * 16 + g * 4 + r;
            if (cin >= C) continue;
#pragma unroll
            for (int nt = 0; nt < OT; ++nt)
                dw[((size_t)tap * C + cin) * c.Cout + co0 + nt * 16 + li] = acc[ti][nt][r];
        }
    }
    if (bias_wave && g == 0) {
#pragma unroll
        for (int nt = 0; nt < OT; ++nt) p.db[soff + co0 + nt * 16 + li] = acc[2][nt][0];
    }
    }
    trace_stamp(5);
    trace_note(6, 8); trace_note(7, sq0 < sqn ? (sqn - 1 - sq0) / sqd + 1 : 0);
}


// Everything the backward pass does with g(b, i): dgrad-horz, dgrad-vert and the weight gradients
// (see bwd_scale_k in wgrad.hip for the grid layout).
struct BwdScaleP {
    ConvP h, v;  WgP w;
    int gyh, gyv, gxh, gxv, gxw, nchw;           // nchw = channel chunks (A + V) of the wgrad
};

// (h, v may be NULL) -> q.h / q.v / q.w, `split` = the weight-gradient split; shape checks of mpnn_msconv_bwd_scale
int mpnn_fill_bwd_scale(const mpnn_dgrad_horz_args *h, const mpnn_dgrad_vert_args *v, const mpnn_wgrad_args *w,
                        BwdScaleP &q, int &split);
