// mpnn_msconv_wgrad: weight / bias gradients of one MultiscaleConvMax scale on
// v_mfma_f32_16x16x4_f32, and mpnn_slab_reduce, the deterministic sum of the
// per-workgroup partial gradients.
//
//   dW_horz[t][c][co] = sum_pix act(a)[pix + t][c]      * g[pix][co]
//   dW_vert[t][c][co] = sum_pix v[pix + t][c] * g[pix][co]      (v = the pooled finer map)
//   db[co]            = sum_pix g[pix][co]
//
// GEMM view per tap: M = 16 input channels (one chunk), N = output channels,
// K = pixels.  grid = (pixel splits, channel chunks of A then V, cout groups).
// A workgroup keeps 9 taps x OT cout tiles of fp32 accumulators in registers
// (waves own taps {w, w+4, w+8}; wave 1's spare slot accumulates the bias
// gradient with a one-hot A operand) while it walks its share of the 64-pixel
// tiles, prefetching the next tile's global loads into registers under the
// MFMAs.  Partial sums leave with plain stores into slab `blockIdx.x`
// (fp32 atomics from hundreds of workgroups onto a few-KB gradient tensor
// serialise: 54 us instead of < 10 for the 16->16 layers); mpnn_slab_reduce
// adds the slabs in a fixed order, so gradients are bitwise reproducible.
//
// LDS: the input chunk's halo tile in the [plane][pixel] float4 layout of the
// forward conv but with plane stride == 1 mod 8 slots so that the 16 channels x
// 2 pixel groups of a ds_read_b32 wave-half fall on 32 distinct banks; the g
// tile as [64 pixels][OT*16 + 4].
#include "bwd_bodies.h"
#include "opt_body.h"

template <int GK, int OT, bool SMALLC = false>
__global__ __launch_bounds__(256) void wgrad_k(const WgP p) {
    constexpr int PS = WGeom<GK>::PS, GS = OT * 16 + 4;
    // one arena: the tile and g buffers are contiguous (the 16-channel body's final reduction uses them as one)
    __shared__ __attribute__((aligned(16))) char smem[4 * PS * 16 + 64 * GS * 4 + (128 * 3 + OT * 16 * 5) * 4];
    f32x4 *tile = (f32x4 *)smem;
    float *gt = (float *)(smem + 4 * PS * 16);
    float *cA = gt + 64 * GS;
    if ((int)blockIdx.y >= ((p.c.a.C + 15) >> 4)) wgrad_body<GK, OT, 1>(p, tile, gt, cA, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x);
    else                                           wgrad_body<GK, OT, 0, SMALLC>(p, tile, gt, cA, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x);
}

// ---------------------------------------------------------------------------
// mpnn_msconv_bwd_scale: everything the backward pass does with g(b, i) -- the input gradients
// through the horz and vert convs and the weight gradients -- as ONE launch.  The three bodies
// are independent (they only share their input), so their workgroups overlap in time instead of
// queueing as three launches with a serial latency chain each.  Grid rows:
//   [0, gyh)            dgrad-horz tiles  (16 output channels each)
//   [gyh, gyh+gyv)      dgrad-vert tiles
//   [gyh+gyv, ...)      wgrad (chunk, 16-cout group) pairs
// ---------------------------------------------------------------------------

#ifndef MPNN_OCC_BWD
#define MPNN_OCC_BWD 3       // waves per SIMD asked of the narrow backward kernel (4 = 128 VGPRs spilled 26-42 registers)
#endif
// HASV = the launch has a dgrad-vert body.  Launches without one (the finest scale of a block) get
// their own instantiation: the dgrad-vert epilogue prefetch costs ~30 registers, and the narrow
// variant without it fits 4 waves per SIMD (1024 resident workgroups instead of 768).
// SMALLC = operand A of the weight gradients is a 1- or 3-channel image (block 0): its own instantiation, with the
// swapped-role weight-gradient body for the image chunk INSTEAD of the general one (both in one kernel spilled).
template <int GK, int OT, int NCH, bool HASV, bool SMALLC = false>
__global__ __launch_bounds__(256, (OT == 1 && NCH == 1 && GK != 2 ? (HASV || GK == 1 ? MPNN_OCC_BWD : MPNN_OCC) : 2)) void bwd_scale_k(const BwdScaleP q) {   // (4x4 maps: few workgroups, no spills at 2 waves)
    constexpr int CB = ConvSmem<GK, 4, 16, NCH>::BYTES;
    constexpr int GS = OT * 16 + 4;
    constexpr int WB = 4 * WGeom<GK>::PS * 16 + 64 * GS * 4 + (128 * 3 + OT * 16 * 5) * 4;
    __shared__ __attribute__((aligned(16))) char smem[CB > WB ? CB : WB];
    // 1-D grid of exactly the workgroups that have work: [0, gyh*gxh) dgrad-horz, then gyv*gxv
    // dgrad-vert, then the weight-gradient workgroups.  (A 2-D grid padded to the widest body launched
    // workgroups that exit at once; with them only 384 of 512 real workgroups were ever co-resident.)
    const int id = blockIdx.x, wh = q.gyh * q.gxh, wv = q.gyv * q.gxv;
    if (id < wh) {
        const int by = id / q.gxh, bx = id - by * q.gxh;
        conv_body<GK, 1, 1, 4, 1, false, EPI_DGH_BN, NCH>(q.h, bx, by, q.gxh, smem);
    } else if (HASV && id < wh + wv) {
        if constexpr (HASV) {
            const int l = id - wh, by = l / q.gxv, bx = l - by * q.gxv;
            conv_body<GK, 1, 1, 4, 1, false, EPI_DGV, NCH>(q.v, bx, by, q.gxv, smem);
        }
    } else {
        const int l = id - wh - wv, r = l / q.gxw, bx = l - r * q.gxw;
        const int chunk = r % q.nchw, bz = r / q.nchw;
        f32x4 *tile = (f32x4 *)smem;
        float *gt = (float *)(smem + 4 * WGeom<GK>::PS * 16);
        float *cA = gt + 64 * GS;
        if (chunk >= ((q.w.c.a.C + 15) >> 4)) wgrad_body<GK, OT, 1>(q.w, tile, gt, cA, bx, chunk, bz, q.gxw);
        else                                   wgrad_body<GK, OT, 0, SMALLC>(q.w, tile, gt, cA, bx, chunk, bz, q.gxw);
    }
}


template <int GK>
static int wgrad_launch(const WgP &p, int split, hipStream_t st) {
    const ConvP &c = p.c;
    const int nch = ((c.a.C + 15) >> 4) + (c.v ? ((c.Cv + 15) >> 4) : 0);
    dim3 block(256);
    if (c.Cout % 64 == 0) {
        hipLaunchKernelGGL((wgrad_k<GK, 4>), dim3(split, nch, c.Cout / 64), block, 0, st, p);
    } else if (c.Cout % 32 == 0) {
        hipLaunchKernelGGL((wgrad_k<GK, 2>), dim3(split, nch, c.Cout / 32), block, 0, st, p);
    } else if (c.Cout % 16 == 0) {
        if (c.a.C <= 3 && MPNN_WG_SMALLC) hipLaunchKernelGGL((wgrad_k<GK, 1, true>), dim3(split, nch, c.Cout / 16), block, 0, st, p);
        else hipLaunchKernelGGL((wgrad_k<GK, 1>), dim3(split, nch, c.Cout / 16), block, 0, st, p);
    } else return MPNN_E_SHAPE;
    MPNN_LAUNCH_CHECK();
    return 0;
}

int mpnn_trace_install_wgrad(void *buf) { return mpnn_trace_install(buf); }

extern "C" int mpnn_wgrad_tiles(int n, int H, int W) {
    if (W >= 16 && (W % 16) == 0 && (H % 4) == 0) return conv_grid_x<0>(n, H, W);
    if (W == 8 && H == 8) return conv_grid_x<1>(n, 8, 8);
    if (W == 4 && H == 4) return conv_grid_x<2>(n, 4, 4);
    return MPNN_E_SHAPE;
}

static int fill_wgrad(const mpnn_wgrad_args *a, WgP &p, int &split);

extern "C" int mpnn_msconv_wgrad(const mpnn_wgrad_args *a, void *stream) {
    if (a && a->n <= 0) return 0;
    WgP p = {};
    int split = 1;
    const int rc = fill_wgrad(a, p, split);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (a->W >= 16) return wgrad_launch<0>(p, split, st);
    if (a->W == 8) return wgrad_launch<1>(p, split, st);
    return wgrad_launch<2>(p, split, st);
}

static int fill_wgrad(const mpnn_wgrad_args *a, WgP &p, int &split) {
    if (!a || !a->a.x || !a->g || !a->dwa || !a->db) return MPNN_E_ARG;
    if (a->v && !a->dwv) return MPNN_E_ARG;
    if (a->a.C > 128 || a->Cv > 128 || (a->Cv & 3)) return MPNN_E_SHAPE;
    if (a->a.C > 4 && (a->a.C & 3)) return MPNN_E_SHAPE;
    const int tiles = mpnn_wgrad_tiles(a->n, a->H, a->W);
    if (tiles < 0) return tiles;
    split = a->n_split < 1 ? 1 : (a->n_split > tiles ? tiles : a->n_split);
    if (split > 1 && a->split_stride <= 0) return MPNN_E_ARG;
    p.c.a = a->a;  p.c.v = a->v;  p.c.Cv = a->v ? a->Cv : 0;
    p.c.n = a->n;  p.c.H = a->H;  p.c.W = a->W;  p.c.Cout = a->Cout;
    p.g = a->g;  p.dwa = a->dwa;  p.dwv = a->dwv;  p.db = a->db;
    p.split_stride = a->split_stride;  p.n_tiles = tiles;
    if (a->g_ctx) {
        if (!a->g_ctx->s || !a->g_ctx->red) return MPNN_E_ARG;
        p.g_on = 1;  p.g_s = a->g_ctx->s;  p.g_bn = a->g_ctx->bn;  p.g_red = a->g_ctx->red;
        p.g_nslot = a->g_ctx->red_nslot < 1 ? 1 : a->g_ctx->red_nslot;
    }
    return 0;
}

int mpnn_fill_dgrad_horz(const mpnn_dgrad_horz_args *a, ConvP &p);     // conv_dgrad.hip
int mpnn_fill_dgrad_vert(const mpnn_dgrad_vert_args *a, ConvP &p);

template <int GK>
static auto bwd_scale_kernel(bool wide, bool deep, bool hasv, bool smallc = false) -> void (*)(const BwdScaleP) {
    if (smallc && !wide && !deep && MPNN_WG_SMALLC) return hasv ? bwd_scale_k<GK, 1, 1, true, true> : bwd_scale_k<GK, 1, 1, false, true>;
    if (hasv) return deep ? (wide ? bwd_scale_k<GK, 4, 2, true> : bwd_scale_k<GK, 1, 2, true>)
                          : (wide ? bwd_scale_k<GK, 4, 1, true> : bwd_scale_k<GK, 1, 1, true>);
    return deep ? (wide ? bwd_scale_k<GK, 4, 2, false> : bwd_scale_k<GK, 1, 2, false>)
                : (wide ? bwd_scale_k<GK, 4, 1, false> : bwd_scale_k<GK, 1, 1, false>);
}

// Resident workgroups of the kernel mpnn_msconv_bwd_scale runs for this shape (the caller sizes
// the weight-gradient split, and with it the slabs, to a share of them).
// 64-channel weight-gradient groups leave room for three workgroups per CU (157 registers), but a third one only
// pays when the input-gradient bodies can use it: with at most one input-gradient workgroup per CU (the 4x4 maps of
// the deep blocks: 32 tiles x 8 rows) it only buys a finer weight-gradient split, i.e. more slabs.
static int wide_cap(bool wide, long dgrad_items) {
    if (!wide) return 0;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    return dgrad_items <= cus ? 2 : 0;
}

extern "C" int mpnn_msconv_bwd_scale_slots(int H, int W, int Cout, int has_dgrad, int has_vert, int dgrad_items) {
    static const int nch_env = [] { const char *e = getenv("MPNN_CONV_NCH"); return e ? atoi(e) : 0; }();
    const bool wide = (Cout % 64) == 0;
    const int gk = (W >= 16 && (W % 16) == 0 && (H % 4) == 0) ? 0 : (W == 8 && H == 8) ? 1 : (W == 4 && H == 4) ? 2 : -1;
    if (gk < 0 || (Cout % 16)) return MPNN_E_SHAPE;
    bool deep = gk != 0 && (Cout % 32) == 0 && has_dgrad;
    if (nch_env != 2) deep = false;        // 32-channel units: opt-in (MPNN_CONV_NCH=2); their 82 KB of LDS leaves ONE workgroup per CU
    const bool hv = has_vert != 0;
    const void *k = gk == 0 ? (const void *)bwd_scale_kernel<0>(wide, deep, hv)
                  : gk == 1 ? (const void *)bwd_scale_kernel<1>(wide, deep, hv) : (const void *)bwd_scale_kernel<2>(wide, deep, hv);
    return resident_slots(k, 0, 256, wide_cap(wide, dgrad_items));
}

template <int GK>
static int bwd_scale_launch(BwdScaleP &q, bool has_h, bool has_v, int split, hipStream_t st) {
    const int tiles = conv_grid_x<GK>(q.w.c.n, q.w.c.H, q.w.c.W);
    q.gyh = has_h ? q.h.Cout / 16 : 0;
    q.gyv = has_v ? q.v.Cout / 16 : 0;
    q.h.n_tiles = q.v.n_tiles = tiles;
    q.h.xcd = q.v.xcd = q.w.c.xcd = xcd_env();
    q.gxw = split;
    q.nchw = ((q.w.c.a.C + 15) >> 4) + (q.w.c.v ? ((q.w.c.Cv + 15) >> 4) : 0);
    const bool wide = (q.w.c.Cout % 64) == 0;           // 64-channel weight-gradient groups for wide layers
    const int gyw = q.nchw * (q.w.c.Cout / (wide ? 64 : 16));
    // 32-channel units for the dgrad bodies when g has a multiple of 32 channels on a small map
    static const int nch_env = [] { const char *e = getenv("MPNN_CONV_NCH"); return e ? atoi(e) : 0; }();
    bool deep = GK != 0 && (q.w.c.Cout % 32) == 0 && (has_h || has_v);
    if (nch_env != 2) deep = false;        // 32-channel units: opt-in (MPNN_CONV_NCH=2); their 82 KB of LDS leaves ONE workgroup per CU
    void (*kern)(const BwdScaleP) = bwd_scale_kernel<GK>(wide, deep, has_v, q.w.c.a.C <= 3);
    // Fit the grid to what is resident at once: the weight-gradient rows keep their split x rows
    // workgroups (the slabs are sized for them), the two dgrad bodies share the rest by work.
    const long slots = resident_slots((const void *)kern, 0, 256, wide_cap(wide, (long)tiles * (q.gyh + q.gyv)));
    long avail = slots - (long)split * gyw;
    if (avail < slots / 4) avail = slots / 4;
    const long units = (q.w.c.Cout + 15) >> 4;          // g's 16-channel chunks: units per dgrad tile
    const long wh = has_h ? (long)q.gyh * units : 0, wv = has_v ? (long)q.gyv * units : 0;
    auto share = [&](long w, int gy) {
        if (!w) return 0;
        long g = avail * w / (wh + wv) / gy;
        if (g < 1) g = 1;
        if (g > tiles) g = tiles;
        return xcd_round((int)g);
    };
    q.gxh = share(wh, q.gyh);
    q.gxv = share(wv, q.gyv);
    const dim3 grid(q.gyh * q.gxh + q.gyv * q.gxv + gyw * q.gxw);
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, st, q);
    MPNN_LAUNCH_CHECK();
    return 0;
}

int mpnn_fill_bwd_scale(const mpnn_dgrad_horz_args *h, const mpnn_dgrad_vert_args *v, const mpnn_wgrad_args *w,
                        BwdScaleP &q, int &split) {
    if (!w) return MPNN_E_ARG;
    int rc = fill_wgrad(w, q.w, split);
    if (rc) return rc;
    if (w->Cout % 16) return MPNN_E_SHAPE;
    if (h) {
        if (!h->prev) return MPNN_E_ARG;
        if ((rc = mpnn_fill_dgrad_horz(h, q.h))) return rc;
        if (h->H != w->H || h->W != w->W || h->n != w->n || (h->Cout % 16) || (h->Cg & 3)) return MPNN_E_ARG;
    }
    if (v) {
        if ((rc = mpnn_fill_dgrad_vert(v, q.v))) return rc;
        if (v->H != w->H || v->W != w->W || v->n != w->n || (v->Cout % 16) || (v->Cg & 3)) return MPNN_E_ARG;
    }
    return 0;
}

extern "C" int mpnn_msconv_bwd_scale(const mpnn_dgrad_horz_args *h, const mpnn_dgrad_vert_args *v,
                                     const mpnn_wgrad_args *w, void *stream) {
    if (!w) return MPNN_E_ARG;
    if (w->n <= 0) return 0;
    BwdScaleP q = {};
    int split = 1;
    const int rc = mpnn_fill_bwd_scale(h, v, w, q, split);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (w->W >= 16 && (w->W % 16) == 0 && (w->H % 4) == 0) return bwd_scale_launch<0>(q, h, v, split, st);
    if (w->W == 8 && w->H == 8) return bwd_scale_launch<1>(q, h, v, split, st);
    if (w->W == 4 && w->H == 4) return bwd_scale_launch<2>(q, h, v, split, st);
    return MPNN_E_SHAPE;
}

// ---------------------------------------------------------------------------
// mpnn_slab_reduce: dst[i] = sum_{s < n_split} src[s * stride + i], in a fixed order.
// table: 6 ints per work item: src_off, dst_off, count (<= MPNN_SLAB_ITEM = 1024), n_split, stride, -.
// One workgroup per item.  Its 256 threads are (element quad q, slab group grp): the item's count / 4
// quads times as many groups G (a power of two, <= 16) as fit; group grp sums slabs grp, grp + G, ...
// with sixteen 16-byte loads in flight, and the groups' partial sums meet in LDS in group order.  The
// caller sizes the items so that n_split / G <= 16 -- ONE memory round trip per workgroup whatever the
// split (1024 elements at a split of 16 or less, 64 elements at 256): few, full workgroups for the big
// tensors of the deep blocks, slab-parallel ones for the small tensors with hundreds of slabs.
// (History: fixed 256-element items, slabs dealt to the four waves: 6 500 workgroups of two loads per
// thread for the chains -- latency, not bandwidth: 2 TB/s.)
// ---------------------------------------------------------------------------
// gl (optional, LDS, >= count floats): the reduced values are ALSO left there (mpnn_backward_finish_opt applies the
// parameter update to them in the same workgroup).
__device__ __forceinline__ void slab_item(const float *__restrict__ slabs, float *__restrict__ grads, const int *__restrict__ t,
                                          float *gl = nullptr) {
    const int src = t[0], dst = t[1], cnt = t[2], ns = t[3], stride = t[4];
    __shared__ f32x4 part[256];
    const int quads = (cnt + 3) >> 2;
    int G = 1;
    while (G < 16 && 2 * G * quads <= 256) G *= 2;              // uniform
    const int tid = threadIdx.x, grp = tid / quads, q = tid - grp * quads;
    const int i = q * 4;
    const bool on_t = grp < G;
    const bool vec = ((src | stride | dst) & 3) == 0;          // uniform
    f32x4 tot = {0.f, 0.f, 0.f, 0.f};
    if (on_t) {
        if (vec && i + 4 <= cnt) {
            const float *base = slabs + src + i;
            f32x4 a[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int sl = grp; sl < ns; sl += 16 * G) {        // (one trip when the caller sized the item)
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int s_ = sl + u * G;
                    const bool on = s_ < ns;
                    const f32x4 v = *(const f32x4 *)(base + (size_t)(on ? s_ : grp) * stride);
                    a[u] += on ? v : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            tot = (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) +
                  (((a[8] + a[9]) + (a[10] + a[11])) + ((a[12] + a[13]) + (a[14] + a[15])));
        } else {
            for (int j = 0; j < 4 && i + j < cnt; ++j) {
                float acc = 0.f;
                for (int sl = grp; sl < ns; sl += G) acc += slabs[src + (size_t)sl * stride + i + j];
                tot[j] = acc;
            }
        }
    }
    if (G > 1) {                                                // uniform
        part[tid] = tot;
        __syncthreads();
        if (grp == 0) {
            for (int k = 1; k < G; ++k) tot += part[k * quads + q];
        }
    }
    if (on_t && grp == 0) {
        if (vec && i + 4 <= cnt) *(f32x4 *)(grads + dst + i) = tot;
        else for (int j = 0; j < 4 && i + j < cnt; ++j) grads[dst + i + j] = tot[j];
        if (gl) for (int j = 0; j < 4 && i + j < cnt; ++j) gl[i + j] = tot[j];
    }
}

__global__ __launch_bounds__(256) void slab_reduce_k(const float *__restrict__ slabs, float *__restrict__ grads,
                                                     const int *__restrict__ table) {
    slab_item(slabs, grads, table + blockIdx.x * 6);
}

// mpnn_backward_finish: the two launches that end the backward pass -- the slab reduction and
// mpnn_bn_finalize -- as one (they are independent of each other).
__global__ __launch_bounds__(256) void backward_finish_k(const float *__restrict__ slabs, float *__restrict__ grads,
                                                         const int *__restrict__ slab_table, int n_items,
                                                         double *__restrict__ sums, double *__restrict__ reds,
                                                         float *__restrict__ state, const int *__restrict__ bn_table,
                                                         float decay, int n_img, double *__restrict__ sums_keep) {
    if ((int)blockIdx.x < n_items) slab_item(slabs, grads, slab_table + blockIdx.x * 6);
    else bn_finalize_body(sums, reds, state, grads, bn_table + (blockIdx.x - n_items) * 8, decay, n_img, sums_keep);
}

extern "C" int mpnn_backward_finish(const float *slabs, float *grads, const int *slab_table, int n_items,
                                    double *sums, double *reds, float *state, const int *bn_table,
                                    int n_bn, float decay, int n_img, double *sums_keep, void *stream) {
    if (n_items < 0 || n_bn < 0 || n_items + n_bn == 0) return n_items + n_bn == 0 ? 0 : MPNN_E_ARG;
    hipLaunchKernelGGL(backward_finish_k, dim3(n_items + n_bn), dim3(256), 0, (hipStream_t)stream, slabs, grads, slab_table,
                       n_items, sums, reds, state, bn_table, decay, n_img, sums_keep);
    MPNN_LAUNCH_CHECK();
    return 0;
}

// mpnn_backward_finish_opt: mpnn_backward_finish AND mpnn_talr_momentum_step as one launch (single-process training:
// nothing sits between the gradients and their use).  Workgroups:
//   [0, n_items)              a slab item: the sum over the slabs, then the update of exactly those elements (the
//                             gradient goes from the reduction to the update through LDS; seg row = item_seg[item])
//   [.., + n_bn)              a BatchNorm: moving averages, dgamma / dbeta, and their update (bn_opt[bn] = node, l2 bits
//                             of gamma, l2 bits of beta, -)
//   [.., + n_plain)           an optimizer work item whose gradient is already final in `grads` (exit parameters,
//                             tensors whose weight gradient was written without slabs)
// Same arithmetic, element for element, as the two launches it replaces.
__device__ __forceinline__ void finish_opt_body(const int b, const float *__restrict__ slabs, const int *__restrict__ slab_table,
                                                int n_items, const int *__restrict__ item_seg,
                                                double *__restrict__ sums, double *__restrict__ reds,
                                                float *__restrict__ state, const int *__restrict__ bn_table, int n_bn,
                                                const int *__restrict__ bn_opt, float decay, int n_img,
                                                double *__restrict__ sums_keep, const OptP &o,
                                                const int *__restrict__ plain_seg) {
    __shared__ float wl[2048];
    __shared__ float gl[MPNN_SLAB_ITEM];
    if (b < n_items) {
        slab_item(slabs, const_cast<float *>(o.grads), slab_table + b * 6, gl);
        __syncthreads();
        opt_seg(o, item_seg + b * MPNN_SEG_INTS, gl, wl);
    } else if (b < n_items + n_bn) {
        const int k = b - n_items;
        const int node = bn_opt[k * 4];
        const float l2g = __int_as_float(bn_opt[k * 4 + 1]), l2b = __int_as_float(bn_opt[k * 4 + 2]);
        float scale, pbar;
        opt_node(o, node, 0, scale, pbar);
        bn_finalize_body(sums, reds, state, const_cast<float *>(o.grads), bn_table + k * 8, decay, n_img, sums_keep,
                         [&](int off_b, float db, int off_g, float dg) {
                             opt_elem(o, off_b, db, l2b, scale, pbar);
                             opt_elem(o, off_g, dg, l2g, scale, pbar);
                         });
    } else {
        opt_seg(o, plain_seg + (b - n_items - n_bn) * MPNN_SEG_INTS, nullptr, wl);
    }
}

__global__ __launch_bounds__(256) void finish_opt_k(const float *__restrict__ slabs, const int *__restrict__ slab_table,
                                                    int n_items, const int *__restrict__ item_seg,
                                                    double *__restrict__ sums, double *__restrict__ reds,
                                                    float *__restrict__ state, const int *__restrict__ bn_table, int n_bn,
                                                    const int *__restrict__ bn_opt, float decay, int n_img,
                                                    double *__restrict__ sums_keep, const OptP o,
                                                    const int *__restrict__ plain_seg) {
    finish_opt_body(blockIdx.x, slabs, slab_table, n_items, item_seg, sums, reds, state, bn_table, n_bn, bn_opt, decay, n_img,
                    sums_keep, o, plain_seg);
}

// Several nets in one launch (co-training, lib/_co.py): wpn workgroups per net (the largest net's count; the others'
// surplus workgroups exit), net r's arguments in the device record tab[r].
__global__ __launch_bounds__(256) void finish_opt_multi_k(const mpnn_finish_net *__restrict__ tab, const int wpn, const float decay) {
    const int net = blockIdx.x / wpn, b = blockIdx.x - net * wpn;
    const mpnn_finish_net f = tab[net];             // (by value: every field's scalar load in the entry block)
    if (b >= f.n_items + f.n_bn + f.n_plain) return;
    const OptP o = {f.params, f.accum, f.grads, f.node_stat, f.hyp, f.talr, f.inv_n, f.grad_scale, f.w_eq, f.packs};
    finish_opt_body(b, f.slabs, f.slab_table, f.n_items, f.item_seg, f.sums, f.reds, f.state, f.bn_table, f.n_bn, f.bn_opt,
                    decay, f.n_img, f.sums_keep, o, f.plain_seg);
}

extern "C" int mpnn_backward_finish_opt(const float *slabs, const int *slab_table, int n_items, const int *item_seg,
                                        double *sums, double *reds, float *state, const int *bn_table, int n_bn,
                                        const int *bn_opt, float decay, int n_img, double *sums_keep,
                                        float *params, float *accum, float *grads, const float *node_stat,
                                        const float *hyp, int talr, float inv_n, float grad_scale, const float *w_eq,
                                        float *packs, const int *plain_seg, int n_plain, void *stream) {
    if (n_items < 0 || n_bn < 0 || n_plain < 0) return MPNN_E_ARG;
    if (n_items + n_bn + n_plain == 0) return 0;
    if ((n_items && (!slabs || !slab_table || !item_seg)) || (n_bn && (!bn_table || !bn_opt)) || (n_plain && !plain_seg) ||
        !params || !accum || !grads || !node_stat || !hyp) return MPNN_E_ARG;
    const OptP o = {params, accum, grads, node_stat, hyp, talr, inv_n, grad_scale, w_eq, packs};
    hipLaunchKernelGGL(finish_opt_k, dim3(n_items + n_bn + n_plain), dim3(256), 0, (hipStream_t)stream, slabs, slab_table,
                       n_items, item_seg, sums, reds, state, bn_table, n_bn, bn_opt, decay, n_img, sums_keep, o, plain_seg);
    MPNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mpnn_backward_finish_opt_multi(const mpnn_finish_net *host_table, const mpnn_finish_net *dev_table, int count,
                                              float decay, void *stream) {
    if (count <= 0) return 0;
    if (!host_table || !dev_table) return MPNN_E_ARG;
    int wpn = 0;
    for (int k = 0; k < count; ++k) {
        const mpnn_finish_net &f = host_table[k];
        if (f.n_items < 0 || f.n_bn < 0 || f.n_plain < 0) return MPNN_E_ARG;
        if ((f.n_items && (!f.slabs || !f.slab_table || !f.item_seg)) || (f.n_bn && (!f.bn_table || !f.bn_opt)) ||
            (f.n_plain && !f.plain_seg) || !f.params || !f.accum || !f.grads || !f.node_stat || !f.hyp) return MPNN_E_ARG;
        const int w = f.n_items + f.n_bn + f.n_plain;
        if (w > wpn) wpn = w;
    }
    if (wpn == 0) return 0;
    hipLaunchKernelGGL(finish_opt_multi_k, dim3(wpn * count), dim3(256), 0, (hipStream_t)stream, dev_table, wpn, decay);
    MPNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mpnn_slab_reduce(const float *slabs, float *grads, const int *table, int n_items, void *stream) {
    if (n_items <= 0) return 0;
    hipLaunchKernelGGL(slab_reduce_k, dim3(n_items), dim3(256), 0, (hipStream_t)stream, slabs, grads, table);
    MPNN_LAUNCH_CHECK();
    return 0;
}
