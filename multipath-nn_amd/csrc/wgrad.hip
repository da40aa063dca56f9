// mpnn_msconv_wgrad: weight / bias gradients of one MultiscaleConvMax scale on
// v_mfma_f32_16x16x4_f32.
//
//   dW_horz[t][c][co] = sum_pix act(a)[pix + t][c]      * g[pix][co]
//   dW_vert[t][c][co] = sum_pix maxpool2(v)[pix + t][c] * g[pix][co]
//   db[co]            = sum_pix g[pix][co]
//
// GEMM view per tap: M = 16 input channels (one chunk), N = output channels,
// K = pixels.  grid = (pixel splits, channel chunks of A then V, cout groups).
// A workgroup keeps 9 taps x OT cout tiles of fp32 accumulators in registers
// (waves own taps {w, w+4, w+8}) while it walks its share of the 64-pixel
// tiles, then adds them to the HWIO gradient with fp32 atomics (bytes of
// atomics = splits x |dW|; the host picks the split).
//
// LDS: the input chunk's halo tile in the same [plane][pixel] float4 layout the
// forward conv stages (stage_chunk is shared) but with plane stride == 1 mod 8
// slots so that the 16 channels x 2 pixel groups of a ds_read_b32 wave-half
// fall on 32 distinct banks; the g tile as [64 pixels][OT*16 + 4].
#include "conv_kernel.h"

struct WgP {
    ConvP c;                 // a, v, Cv, n, H, W used by stage_chunk
    const float *g;
    float *dwa, *dwv, *db;
    int n_tiles;
};

template <int GK> struct WGeom;
template <> struct WGeom<0> { static constexpr int PS = 113; };
template <> struct WGeom<1> { static constexpr int PS = 241; };
template <> struct WGeom<2> { static constexpr int PS = 289; };

template <int GK, int PS, int MODE>
__device__ __forceinline__ void stage_chunk_ps(f32x4 *tile, const ConvP &p, const float *cA,
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
                        v[k] = (c + k < C) ? fmaxf((v[k] - cc[0]) * cc[1] + cc[2], 0.f) : 0.f;
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
        tile[q * PS + (img * HR + hy) * G::R + hx] = v;
    }
}

template <int GK, int OT>
__global__ __launch_bounds__(256) void wgrad_k(const WgP p) {
    using G = Geom<GK>;
    constexpr int PS = WGeom<GK>::PS, R = G::R, HR = G::TH + 2;
    constexpr int GS = OT * 16 + 4;                  // g tile row stride (floats)
    __shared__ f32x4 tile[4 * PS];
    __shared__ float gt[64 * GS];
    __shared__ float cA[128 * 3];

    const ConvP &c = p.c;
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    const int g = lane >> 4, li = lane & 15;
    const int nchA = (c.a.C + 15) >> 4;
    const int part = (int)blockIdx.y >= nchA ? 1 : 0;
    const int ch = part ? (int)blockIdx.y - nchA : (int)blockIdx.y;
    const int C = part ? c.Cv : c.a.C;
    const int co0 = blockIdx.z * OT * 16;
    int np = (C - ch * 16 + 3) >> 2;
    np = np > 4 ? 4 : np;

    if (part == 0 && c.a.mode != MPNN_ACT_IDENTITY) {
        for (int cc = tid; cc < c.a.C; cc += 256) {
            const BnC k = bn_coef(c.a, cc);
            cA[cc * 3] = k.m; cA[cc * 3 + 1] = k.gamma * k.rstd; cA[cc * 3 + 2] = k.beta;
        }
    }

    f32x4 acc[3][OT];
#pragma unroll
    for (int ti = 0; ti < 3; ++ti)
#pragma unroll
        for (int nt = 0; nt < OT; ++nt) acc[ti][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbacc = 0.f;
    const float *tf = (const float *)tile;
    const int a_lane = (li >> 2) * PS * 4 + (li & 3);      // plane + component of channel li

    for (int t = blockIdx.x; t < p.n_tiles; t += gridDim.x) {
        int n0, y0, x0;
        tile_origin<GK>(c, t, n0, y0, x0);
        __syncthreads();
        if (part == 0) stage_chunk_ps<GK, PS, 0>(tile, c, cA, n0, y0, x0, ch * 16, np, tid);
        else           stage_chunk_ps<GK, PS, 1>(tile, c, cA, n0, y0, x0, ch * 16, np, tid);
        for (int i = tid; i < 64 * OT * 4; i += 256) {
            const int q = i % (OT * 4), pi = i / (OT * 4);
            int img, ty, tx;
            mtile_pix<GK>(pi >> 4, pi & 15, img, ty, tx);
            const int n = n0 + img;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (n < c.n)
                v = *(const f32x4 *)(p.g + (((size_t)n * c.H + y0 + ty) * c.W + x0 + tx) * c.Cout + co0 + q * 4);
            *(f32x4 *)(gt + pi * GS + q * 4) = v;
        }
        __syncthreads();
        if (blockIdx.y == 0 && tid < OT * 16) {
            float s = 0.f;
#pragma unroll 8
            for (int pi = 0; pi < 64; ++pi) s += gt[pi * GS + tid];
            dbacc += s;
        }
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int img, ty, tx;
                mtile_pix<GK>(kc, 4 * g + j, img, ty, tx);
                const int slot = (img * HR + ty) * R + tx;
                float b[OT];
#pragma unroll
                for (int nt = 0; nt < OT; ++nt) b[nt] = gt[(kc * 16 + 4 * g + j) * GS + nt * 16 + li];
#pragma unroll
                for (int ti = 0; ti < 3; ++ti) {
                    const int tap = wid + 4 * ti;
                    if (tap < 9) {
                        const int dy = tap / 3, dx = tap - dy * 3;
                        const float a = tf[(slot + dy * R + dx) * 4 + a_lane];
#pragma unroll
                        for (int nt = 0; nt < OT; ++nt)
                            acc[ti][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[nt], acc[ti][nt], 0, 0, 0);
                    }
                }
            }
        }
    }

    // D layout: col = li (cout), row = g*4 + r (input channel of the chunk).
    float *dw = part ? p.dwv : p.dwa;
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
                atomicAdd(dw + ((size_t)tap * C + cin) * c.Cout + co0 + nt * 16 + li, acc[ti][nt][r]);
        }
    }
    if (blockIdx.y == 0 && tid < OT * 16) atomicAdd(p.db + co0 + tid, dbacc);
}

template <int GK>
static int wgrad_launch(const WgP &p, int n_split, hipStream_t st) {
    const ConvP &c = p.c;
    const int nch = ((c.a.C + 15) >> 4) + (c.v ? ((c.Cv + 15) >> 4) : 0);
    int split = n_split < 1 ? 1 : (n_split > p.n_tiles ? p.n_tiles : n_split);
    dim3 block(256);
    if (c.Cout % 64 == 0) {
        hipLaunchKernelGGL((wgrad_k<GK, 4>), dim3(split, nch, c.Cout / 64), block, 0, st, p);
    } else if (c.Cout % 32 == 0) {
        hipLaunchKernelGGL((wgrad_k<GK, 2>), dim3(split, nch, c.Cout / 32), block, 0, st, p);
    } else if (c.Cout % 16 == 0) {
        hipLaunchKernelGGL((wgrad_k<GK, 1>), dim3(split, nch, c.Cout / 16), block, 0, st, p);
    } else return MPNN_E_SHAPE;
    MPNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mpnn_msconv_wgrad(const mpnn_wgrad_args *a, void *stream) {
    if (!a || !a->a.x || !a->g || !a->dwa || !a->db) return MPNN_E_ARG;
    if (a->v && !a->dwv) return MPNN_E_ARG;
    if (a->n <= 0) return 0;
    if (a->a.C > 128 || a->Cv > 128 || (a->Cv & 3)) return MPNN_E_SHAPE;
    if (a->a.C > 4 && (a->a.C & 3)) return MPNN_E_SHAPE;
    WgP p = {};
    p.c.a = a->a;  p.c.v = a->v;  p.c.Cv = a->v ? a->Cv : 0;
    p.c.n = a->n;  p.c.H = a->H;  p.c.W = a->W;  p.c.Cout = a->Cout;
    p.g = a->g;  p.dwa = a->dwa;  p.dwv = a->dwv;  p.db = a->db;
    hipStream_t st = (hipStream_t)stream;
    if (a->W >= 16 && (a->W % 16) == 0 && (a->H % 4) == 0) {
        p.n_tiles = conv_grid_x<0>(a->n, a->H, a->W);
        return wgrad_launch<0>(p, a->n_split, st);
    }
    if (a->W == 8 && a->H == 8) { p.n_tiles = conv_grid_x<1>(a->n, 8, 8); return wgrad_launch<1>(p, a->n_split, st); }
    if (a->W == 4 && a->H == 4) { p.n_tiles = conv_grid_x<2>(a->n, 4, 4); return wgrad_launch<2>(p, a->n_split, st); }
    return MPNN_E_SHAPE;
}
