// Direct (im2col-free) 3x3 SAME convolution on v_mfma_f32_16x16x4_f32 for gfx950.
//
// One template serves the forward conv of a MultiscaleConvMax scale and both
// input-gradient convs (they are the same contraction with a transposed,
// tap-flipped weight pack); what differs is the operand transform applied while
// staging and the epilogue:
//
//   EPI_FWD      out = bias + conv(act(a)) [+ conv(maxpool2(v))], BN sum/sumsq
//   EPI_DGH_BN   dz = [y>0] * (conv^T(g) [+ extra]),  BN-backward reductions
//   EPI_DGH_RAW  dy = conv^T(g) [+ extra]
//   EPI_DGV      g_fine = bn_bwd(dz_fine) + maxpool_bwd(conv^T(g))
//
// Implicit GEMM: M = output pixels, N = output channels, K = 9 taps x input
// channels.  A workgroup (4 waves) owns a tile of 64 output pixels x CT output
// channels.  Input channels are processed in chunks of 16: the chunk's halo
// tile is staged into LDS as 4 "planes" of float4 (4 consecutive channels of
// one pixel per 16-byte slot, layout [plane][halo pixel]), so that ONE
// ds_read_b128 gives a lane the A operands of four consecutive MFMAs (MFMA j
// of the group contracts channels {4g+j}, g = lane>>4), and the k-interleaved
// weight pack gives the matching B operands with ONE global_load_dwordx4
// (L2-resident, no LDS staging, no barrier inside the 9-tap loop).
//
// Bank conflicts: ds_read_b128 is served in 16-lane groups that mix lanes of
// two g values; with the plane stride P a multiple of 16 slots and the 16
// pixels of an M-tile on distinct slots mod 16 (row stride R chosen per
// geometry) every read is conflict-free; staging writes put 8 consecutive
// pixels of one plane in the 8 lanes of a ds_write_b128 group.
#pragma once
#include "common.h"

enum { EPI_FWD = 0, EPI_DGH_BN = 1, EPI_DGH_RAW = 2, EPI_DGV = 3 };

struct ConvP {
    mpnn_act a;                 // operand A (identity transform for dgrad)
    const float *v;  int Cv;    // operand V: finer pre-BN map, pooled on load
    const float *wa, *wv;       // weight packs
    int n, H, W, Cout;
    const float *bias;  float *out;  double *out_sum;      // EPI_FWD
    const float *extra;                                     // EPI_DGH_*
    const float *sprev;  mpnn_act pbn;  double *red_out;    // EPI_DGH_BN / EPI_DGV
    const double *red;  int has_dz;                         // EPI_DGV
};

// Geometry kinds.  TH x TW output pixels per image x IMG images = 64 pixels;
// an M-tile is 16 of them.  R = LDS row stride (slots), P = plane stride.
template <int GK> struct Geom;
template <> struct Geom<0> { static constexpr int TH = 4, TW = 16, IMG = 1, R = 18, P = 112; };  // W % 16 == 0
template <> struct Geom<1> { static constexpr int TH = 8, TW = 8,  IMG = 1, R = 24, P = 240; };  // 8 x 8 maps
template <> struct Geom<2> { static constexpr int TH = 4, TW = 4,  IMG = 4, R = 12, P = 288; };  // 4 x 4 maps

template <int GK>
__device__ __forceinline__ void mtile_pix(int m, int i, int &img, int &ty, int &tx) {
    if (GK == 0) { img = 0; ty = m; tx = i; }
    else if (GK == 1) { img = 0; ty = 2 * m + (i >> 3); tx = i & 7; }
    else { img = m; ty = i >> 2; tx = i & 3; }
}

template <int GK>
__device__ __forceinline__ void tile_origin(const ConvP &p, int bid, int &n0, int &y0, int &x0) {
    if (GK == 0) {
        const int tx_n = p.W >> 4, tpi = tx_n * (p.H >> 2);
        n0 = bid / tpi;
        const int rem = bid - n0 * tpi;
        const int ty = rem / tx_n;
        y0 = ty * 4; x0 = (rem - ty * tx_n) * 16;
    } else if (GK == 1) { n0 = bid; y0 = 0; x0 = 0; }
    else { n0 = bid * 4; y0 = 0; x0 = 0; }
}

template <int GK>
static inline int conv_grid_x(int n, int H, int W) {
    if (GK == 0) return n * (W >> 4) * (H >> 2);
    if (GK == 1) return n;
    return (n + 3) >> 2;
}

// Stage one 16-channel chunk of the halo tile.  MODE 0: operand A (optional
// BN+ReLU, optional pyramid subsampling); MODE 1: 2x2 max-pool of the finer map.
template <int GK, int MODE>
__device__ __forceinline__ void stage_chunk(f32x4 *tile, const ConvP &p, const float *cA,
                                            int n0, int y0, int x0, int c0, int np, int tid) {
    using G = Geom<GK>;
    constexpr int HR = G::TH + 2, HC = G::TW + 2, NHP = G::IMG * HR * HC, NHP8 = (NHP + 7) & ~7;
    for (int i = tid; i < NHP8 * 4; i += 256) {
        const int q = (i >> 3) & 3;
        const int hp = ((i >> 5) << 3) + (i & 7);
        if (hp >= NHP) continue;
        const int img = hp / (HR * HC);
        const int rem = hp - img * (HR * HC);
        const int hy = rem / HC, hx = rem - hy * HC;
        const int n = n0 + img, y = y0 + hy - 1, x = x0 + hx - 1;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (q < np && n < p.n && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) {
            const int c = c0 + q * 4;
            if (MODE == 0) {
                const int sh = p.a.shift, C = p.a.C;
                const size_t base = (((size_t)n * (p.H << sh) + (y << sh)) * (p.W << sh) + (x << sh)) * C;
                if ((C & 3) == 0) {
                    v = *(const f32x4 *)(p.a.x + base + c);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = (c + k < C) ? p.a.x[base + c + k] : 0.f;
                }
                if (p.a.mode != MPNN_ACT_IDENTITY) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float *cc = cA + (c + k) * 3;
                        v[k] = fmaxf((v[k] - cc[0]) * cc[1] + cc[2], 0.f);
                    }
                }
            } else {
                const int W2 = p.W * 2;
                const float *s = p.v + (((size_t)n * (p.H * 2) + 2 * y) * W2 + 2 * x) * p.Cv + c;
                const f32x4 a0 = *(const f32x4 *)s, a1 = *(const f32x4 *)(s + p.Cv);
                const f32x4 a2 = *(const f32x4 *)(s + (size_t)W2 * p.Cv);
                const f32x4 a3 = *(const f32x4 *)(s + (size_t)W2 * p.Cv + p.Cv);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = fmaxf(fmaxf(a0[k], a1[k]), fmaxf(a2[k], a3[k]));
            }
        }
        tile[q * G::P + (img * HR + hy) * G::R + hx] = v;
    }
}

template <int GK, int MT, int NT, int WM, int WN, bool SMALL_A, int EPI>
__global__ __launch_bounds__(256) void conv_k(const ConvP p) {
    using G = Geom<GK>;
    constexpr int P = G::P, R = G::R, HR = G::TH + 2;
    constexpr int CT = WN * NT * 16;
    static_assert(WM * WN == 4 && WM * MT == 4, "4 waves, 4 M-tiles per workgroup");

    __shared__ f32x4 tile[4 * P];
    __shared__ float cA[128 * 3];
    __shared__ float cE[CT * 5];
    __shared__ double redbuf[WM * CT * 2];

    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    const int g = lane >> 4, li = lane & 15;
    const int wm = wid / WN, wn = wid - wm * WN;
    int n0, y0, x0;
    tile_origin<GK>(p, blockIdx.x, n0, y0, x0);
    const int co0 = blockIdx.y * CT;
    const int cw = co0 + wn * NT * 16 + li;          // this lane's first output channel

    if (EPI == EPI_FWD && p.a.mode != MPNN_ACT_IDENTITY) {
        for (int c = tid; c < p.a.C; c += 256) {
            const BnC k = bn_coef(p.a, c);
            cA[c * 3] = k.m; cA[c * 3 + 1] = k.gamma * k.rstd; cA[c * 3 + 2] = k.beta;
        }
    }
    if (EPI == EPI_DGH_BN || EPI == EPI_DGV) {
        for (int c = tid; c < CT; c += 256) {
            const BnC k = bn_coef(p.pbn, co0 + c);
            float *e = cE + c * 5;
            e[0] = k.m; e[1] = k.rstd; e[2] = k.gamma * k.rstd;
            if (EPI == EPI_DGH_BN) { e[3] = k.beta; e[4] = 0.f; }
            else {
                const double inv = 1.0 / (double)p.pbn.cnt;
                e[3] = p.red ? (float)(slot_sum(p.red, 2 * p.pbn.C, co0 + c) * inv) : 0.f;             // dbeta / cnt
                e[4] = p.red ? (float)(slot_sum(p.red, 2 * p.pbn.C, p.pbn.C + co0 + c) * inv) : 0.f;   // dgamma / cnt
            }
        }
    }

    int slot0[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int img, ty, tx;
        mtile_pix<GK>(wm * MT + mt, li, img, ty, tx);
        slot0[mt] = (img * HR + ty) * R + tx;
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int part = 0; part < 2; ++part) {
        const int C = part == 0 ? p.a.C : (p.v ? p.Cv : 0);
        if (C == 0) continue;
        const float *wp = part == 0 ? p.wa : p.wv;
        const int nch = (C + 15) >> 4;
        for (int ch = 0; ch < nch; ++ch) {
            __syncthreads();
            int np = (C - ch * 16 + 3) >> 2;
            np = np > 4 ? 4 : np;
            if (part == 0) stage_chunk<GK, 0>(tile, p, cA, n0, y0, x0, ch * 16, np, tid);
            else           stage_chunk<GK, 1>(tile, p, cA, n0, y0, x0, ch * 16, np, tid);
            __syncthreads();
            if (SMALL_A && part == 0) {
                const float *tf = (const float *)tile;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int dy = tap / 3, dx = tap - dy * 3;
                    float b[NT], a[MT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        b[nt] = wp[((size_t)(tap * 4) * p.Cout + cw + nt * 16) * 4 + g];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) a[mt] = tf[(slot0[mt] + dy * R + dx) * 4 + g];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int dy = tap / 3, dx = tap - dy * 3;
                    const float *wt = wp + ((size_t)((tap * nch + ch) * 4 + g) * p.Cout + cw) * 4;
                    f32x4 b[NT], a[MT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) b[nt] = *(const f32x4 *)(wt + nt * 64);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) a[mt] = tile[g * P + slot0[mt] + dy * R + dx];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][j], b[nt][j], acc[mt][nt], 0, 0, 0);
                }
            }
        }
    }

    // ------------------------------- epilogue --------------------------------
    // D layout: col = lane & 15 (channel), row = (lane >> 4) * 4 + r (pixel of the M-tile).
    float s1[NT], s2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { s1[nt] = 0.f; s2[nt] = 0.f; }

#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int img, ty, tx;
            mtile_pix<GK>(wm * MT + mt, g * 4 + r, img, ty, tx);
            const int n = n0 + img, y = y0 + ty, x = x0 + tx;
            if (n >= p.n) continue;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = cw + nt * 16;
                const int cl = co - co0;
                float val = acc[mt][nt][r];
                if (EPI == EPI_FWD) {
                    const size_t idx = (((size_t)n * p.H + y) * p.W + x) * p.Cout + co;
                    val += p.bias[co];
                    p.out[idx] = val;
                    s1[nt] += val; s2[nt] += val * val;
                } else if (EPI == EPI_DGH_RAW) {
                    const size_t idx = (((size_t)n * p.H + y) * p.W + x) * p.Cout + co;
                    if (p.extra) val += p.extra[idx];
                    p.out[idx] = val;
                } else if (EPI == EPI_DGH_BN) {
                    const size_t idx = (((size_t)n * p.H + y) * p.W + x) * p.Cout + co;
                    if (p.extra) val += p.extra[idx];
                    const float *e = cE + cl * 5;
                    const float d = p.sprev[idx] - e[0];
                    const float yv = d * e[2] + e[3];
                    const float dz = yv > 0.f ? val : 0.f;
                    p.out[idx] = dz;
                    s1[nt] += dz; s2[nt] += dz * (d * e[1]);
                } else {  // EPI_DGV: val = d(pooled fine map) at coarse pixel (y, x)
                    const float *e = cE + cl * 5;
                    const int W2 = p.W * 2;
                    const size_t i00 = (((size_t)n * (p.H * 2) + 2 * y) * W2 + 2 * x) * p.Cout + co;
                    const size_t ix[4] = {i00, i00 + p.Cout, i00 + (size_t)W2 * p.Cout,
                                          i00 + (size_t)W2 * p.Cout + p.Cout};
                    float sv[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) sv[k] = p.sprev[ix[k]];
                    int arg = 0; float mx = sv[0];
#pragma unroll
                    for (int k = 1; k < 4; ++k) if (sv[k] > mx) { mx = sv[k]; arg = k; }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float dzf = p.has_dz ? p.out[ix[k]] : 0.f;
                        const float xh = (sv[k] - e[0]) * e[1];
                        float gk = e[2] * (dzf - e[3] - xh * e[4]);
                        if (k == arg) gk += val;
                        p.out[ix[k]] = gk;
                    }
                }
            }
        }
    }

    if (EPI == EPI_FWD || EPI == EPI_DGH_BN) {
        double *dst = EPI == EPI_FWD ? p.out_sum : p.red_out;
        if (dst) {
            __syncthreads();     // (redbuf is separate from tile, but keep phases ordered)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const double a1 = reduce_g4((double)s1[nt]);
                const double a2 = reduce_g4((double)s2[nt]);
                if (g == 0) {
                    const int cl = wn * NT * 16 + nt * 16 + li;
                    redbuf[(wm * CT + cl) * 2] = a1;
                    redbuf[(wm * CT + cl) * 2 + 1] = a2;
                }
            }
            __syncthreads();
            if (tid < CT) {
                double a1 = 0.0, a2 = 0.0;
#pragma unroll
                for (int w = 0; w < WM; ++w) { a1 += redbuf[(w * CT + tid) * 2]; a2 += redbuf[(w * CT + tid) * 2 + 1]; }
                double *slot = dst + (size_t)(blockIdx.x % MPNN_BN_SLOTS) * 2 * p.Cout;
                atomicAdd(slot + co0 + tid, a1);
                atomicAdd(slot + p.Cout + co0 + tid, a2);
            }
        }
    }
}

// ------------------------------- host dispatch -------------------------------
template <int GK, int MT, int NT, int WM, int WN, int EPI>
static int conv_launch_cfg(const ConvP &p, bool small_a, hipStream_t st) {
    constexpr int CT = WN * NT * 16;
    dim3 grid(conv_grid_x<GK>(p.n, p.H, p.W), p.Cout / CT), block(256);
    if (EPI == EPI_FWD && small_a)
        hipLaunchKernelGGL((conv_k<GK, MT, NT, WM, WN, true, EPI>), grid, block, 0, st, p);
    else
        hipLaunchKernelGGL((conv_k<GK, MT, NT, WM, WN, false, EPI>), grid, block, 0, st, p);
    MPNN_LAUNCH_CHECK();
    return 0;
}

template <int GK, int EPI>
static int conv_launch_geom(const ConvP &p, bool small_a, hipStream_t st) {
    const int Co = p.Cout;
    if (Co % 16) return MPNN_E_SHAPE;
    // 4x4 maps have few spatial tiles: prefer narrow channel tiles there.
    if (Co % 64 == 0 && GK != 2) return conv_launch_cfg<GK, 2, 2, 2, 2, EPI>(p, small_a, st);
    if (Co % 32 == 0) return conv_launch_cfg<GK, 2, 1, 2, 2, EPI>(p, small_a, st);
    return conv_launch_cfg<GK, 1, 1, 4, 1, EPI>(p, small_a, st);
}

template <int EPI>
static int conv_launch(const ConvP &p, hipStream_t st) {
    if (p.n <= 0) return 0;
    if (p.a.C > 128 || p.Cv > 128 || (p.Cv & 3)) return MPNN_E_SHAPE;
    const bool small_a = p.a.C <= 4;
    if (!small_a && (p.a.C & 3)) return MPNN_E_SHAPE;
    if (p.W >= 16 && (p.W % 16) == 0 && (p.H % 4) == 0) return conv_launch_geom<0, EPI>(p, small_a, st);
    if (p.W == 8 && p.H == 8) return conv_launch_geom<1, EPI>(p, small_a, st);
    if (p.W == 4 && p.H == 4) return conv_launch_geom<2, EPI>(p, small_a, st);
    return MPNN_E_SHAPE;
}
