// mpnn_msconv_fwd: one scale of MultiscaleConvMax forward (see conv_kernel.h).
#include "conv_kernel.h"
#include "conv_strip.h"

__host__ __device__ static inline int fill_fwd(const mpnn_conv_fwd_args *a, ConvP &p);

int mpnn_trace_install_fwd(void *buf) { return mpnn_trace_install(buf); }

extern "C" int mpnn_msconv_fwd(const mpnn_conv_fwd_args *a, void *stream) {
    ConvP p = {};
    const int rc = fill_fwd(a, p);
    if (rc) return rc;
    if (a->idx || a->cnt) return MPNN_E_ARG;          // index lists: mpnn_msconv_fwd_group (device-side records)
    return conv_launch<EPI_FWD>(p, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// mpnn_msconv_fwd_group: up to four forward convs that do not depend on each other (one wavefront
// level of the block x scale grid: F(b, k) needs only F(b-1, k) and F(b, k-1)) as ONE launch.
// Rows of the grid are dealt to the members; each member runs the body of its own geometry.
// The members' serial latency chains overlap instead of queueing as separate launches.
// ---------------------------------------------------------------------------
struct FwdGroupP { int gk[4], small[4], gy[4], gx[4], y0[4], w0[4], rh[4]; int n; int xcd; int reps, wpr; };   // w0 = first linear workgroup of a member; gk 3 = strip body, rh rows per strip
// reps > 1 (mpnn_msconv_fwd_group_rep: co-trained nets of one architecture): the grid is `reps` copies of the group's
// wpr workgroups, copy r runs the records tab[r * n .. r * n + n) -- same geometry, the buffers of net r.

__host__ __device__ static inline int fill_fwd(const mpnn_conv_fwd_args *a, ConvP &p) {
    if (!a || !a->a.x || !a->wa_pack || !a->out || !a->bias) return MPNN_E_ARG;
    if (a->v && !a->wv_pack) return MPNN_E_ARG;
    p.a = a->a;
    p.v = a->v;  p.Cv = a->v ? a->Cv : 0;
    p.wa = a->wa_pack;  p.wv = a->wv_pack;
    p.n = a->n;  p.H = a->H;  p.W = a->W;  p.Cout = a->Cout;
    p.bias = a->bias;  p.out = a->out;  p.out_sum = a->out_sum;  p.pool_out = a->pool_out;
    p.out_nslot = a->out_nslot < 1 ? 1 : (a->out_nslot > MPNN_BN_SLOTS ? MPNN_BN_SLOTS : a->out_nslot);
    if (a->pool_out && (a->H < 8 || (a->H & 1) || (a->W & 1))) return MPNN_E_SHAPE;
    p.idx = a->idx;
    if (a->idx && !a->cnt) return MPNN_E_ARG;
#if defined(__HIP_DEVICE_COMPILE__)
    if (a->cnt) { const int c = *a->cnt; p.n = c < a->n ? c : a->n; }     // device-side count of the routed sub-batch
#endif
    return 0;
}

// The member records live in DEVICE memory (uploaded once per plan): indexing a by-value kernel
// argument array with a runtime member index makes hipcc copy the whole argument block to scratch.
// SMALL = some member has a 1- or 3-channel operand A (block 0).  Groups without one run an
// instantiation that omits those bodies: 128 instead of 160 VGPRs, i.e. 4 instead of 3 waves per SIMD.
// IDX = the members carry index lists (routed evaluation): the same bodies with the image indirection.
// WIDE = every member is an 8x8 / 4x4 conv with Cout % 32 == 0 at an evaluation batch: 32 output channels per
// workgroup (two M-tiles per wave, two wave columns) -- the input tile is staged once per 32 instead of per 16
// output channels; same contraction order per output element, i.e. the same bits as the 16-channel tile.
template <bool SMALL, bool IDX, bool WIDE = false>
__global__ __launch_bounds__(256, WIDE ? 2 : (SMALL ? 3 : MPNN_OCC)) void fwd_group_k(const mpnn_conv_fwd_args *__restrict__ tab, const FwdGroupP q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // sized by the host for the members present
    // 1-D grid with exactly the workgroups that have work (a 2-D grid padded to the widest member
    // launches workgroups that exit at once, and they were seen to delay the residency of real ones)
    int id = blockIdx.x;
    if (q.reps > 1) { const int rep = id / q.wpr; id -= rep * q.wpr; tab += rep * q.n; }     // (uniform)
    int m = 0, w0 = 0, gx = q.gx[0], kind = q.gk[0] * 2 + q.small[0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < q.n && id >= q.w0[k]) { m = k; w0 = q.w0[k]; gx = q.gx[k]; kind = q.gk[k] * 2 + q.small[k]; }
    const int yy = (id - w0) / gx, bx = (id - w0) - yy * gx;
    ConvP p = {};
    fill_fwd(tab + m, p);
    p.xcd = q.xcd;
    if constexpr (WIDE) {
        if (kind == 2) { p.n_tiles = conv_grid_x<1>(p.n, p.H, p.W); conv_body<1, 2, 1, 2, 2, false, EPI_FWD, 1, false, IDX>(p, bx, yy, gx, smem); }
        else           { p.n_tiles = conv_grid_x<2>(p.n, p.H, p.W); conv_body<2, 2, 1, 2, 2, false, EPI_FWD, 1, false, IDX>(p, bx, yy, gx, smem); }
        return;
    }
    switch (kind) {
        case 0: p.n_tiles = conv_grid_x<0>(p.n, p.H, p.W); conv_body<0, 1, 1, 4, 1, false, EPI_FWD, 1, false, IDX>(p, bx, yy, gx, smem); break;
        case 2: p.n_tiles = conv_grid_x<1>(p.n, p.H, p.W); conv_body<1, 1, 1, 4, 1, false, EPI_FWD, 1, false, IDX>(p, bx, yy, gx, smem); break;
        case 4: p.n_tiles = conv_grid_x<2>(p.n, p.H, p.W); conv_body<2, 1, 1, 4, 1, false, EPI_FWD, 1, false, IDX>(p, bx, yy, gx, smem); break;
        case 6: strip16_body<IDX>(tab[m], bx, yy, gx, q.rh[m], q.xcd, smem); break;        // 16 -> 16 k channels on a big map (conv_strip.h)
        case 8: stripk_body<IDX>(tab[m], bx, yy, gx, q.rh[m], q.xcd, smem); break;         // the same with 2-3 input chunks
        default:
            if constexpr (SMALL) {
                if (kind == 9)      stripk_body<IDX, true>(tab[m], bx, yy, gx, q.rh[m], q.xcd, smem);        // image + V on a big map (conv_strip.h)
                else if (kind == 1) { p.n_tiles = conv_grid_x<0>(p.n, p.H, p.W); conv_body<0, 1, 1, 4, 1, true, EPI_FWD, 1, false, IDX>(p, bx, yy, gx, smem); }
                else if (kind == 3) { p.n_tiles = conv_grid_x<1>(p.n, p.H, p.W); conv_body<1, 1, 1, 4, 1, true, EPI_FWD, 1, false, IDX>(p, bx, yy, gx, smem); }
                else                { p.n_tiles = conv_grid_x<2>(p.n, p.H, p.W); conv_body<2, 1, 1, 4, 1, true, EPI_FWD, 1, false, IDX>(p, bx, yy, gx, smem); }
            }
            break;
    }
}

// One deep small-map member alone in its launch: K-split body (see conv_body) -- KS thread groups of 256, each one of the
// unit's KS 16-channel chunks: 512 threads / 32-channel units, or (inputs of >= 128 channels, all chunk counts multiples of
// four) 1 024 threads / 64-channel units: one workgroup per CU, four waves per SIMD, half the unit chain again.
template <int GK, int KS = 2>
__global__ __launch_bounds__(KS * 256) void fwd_ks_k(const mpnn_conv_fwd_args *__restrict__ tab, const int gx, const int xcd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // ConvSmem<GK, 4, 16, KS>::BYTES (above 64 KB for KS = 4)
    const int id = blockIdx.x, yy = id / gx, bx = id - yy * gx;
    ConvP p = {};
    fill_fwd(tab, p);
    p.xcd = xcd;
    p.n_tiles = conv_grid_x<GK>(p.n, p.H, p.W);
    conv_body<GK, 1, 1, 4, 1, false, EPI_FWD, KS, true>(p, bx, yy, gx, smem);
}

int mpnn_first_conv_launch_rep(const mpnn_conv_fwd_args *a, const mpnn_conv_fwd_args *dev_args, int reps, int share, hipStream_t st);

static int fwd_group_launch(const mpnn_conv_fwd_args *args, const mpnn_conv_fwd_args *dev_args, int count, int reps, int share, void *stream);

extern "C" int mpnn_msconv_fwd_group(const mpnn_conv_fwd_args *args, const mpnn_conv_fwd_args *dev_args, int count,
                                     void *stream) {
    return fwd_group_launch(args, dev_args, count, 1, 1, stream);
}

// The same group for `reps` nets of one architecture in ONE launch (co-training, lib/_co.py): args / dev_args hold
// reps * count records, net r's at [r * count, (r + 1) * count) -- identical shapes and modes, the buffers of net r.
// The resident slots are shared between the nets; every net gets the grid the group would get on slots / share
// (share <= 0: reps.  reps = 1 with share = K is the launch of ONE of K nets that take turns with the same grids --
// the same per-workgroup tile runs, i.e. the same fp32 partial sums of the BatchNorm statistics, as in the joint launch).
extern "C" int mpnn_msconv_fwd_group_rep(const mpnn_conv_fwd_args *args, const mpnn_conv_fwd_args *dev_args, int count,
                                         int reps, int share, void *stream) {
    if (reps < 1) return MPNN_E_ARG;
    if (share <= 0) share = reps;
    if (count <= 0) return 0;
    if (!args || count > 4) return MPNN_E_ARG;
    for (int r = 1; r < reps; ++r)
        for (int k = 0; k < count; ++k) {
            const mpnn_conv_fwd_args &a = args[k], &b = args[r * count + k];
            if (a.n != b.n || a.H != b.H || a.W != b.W || a.Cout != b.Cout || a.a.C != b.a.C || a.a.mode != b.a.mode ||
                a.a.shift != b.a.shift || (a.v != nullptr) != (b.v != nullptr) || a.Cv != b.Cv || a.out_nslot != b.out_nslot ||
                (a.pool_out != nullptr) != (b.pool_out != nullptr) || (a.out_sum != nullptr) != (b.out_sum != nullptr) ||
                a.idx || b.idx || a.cnt || b.cnt) return MPNN_E_ARG;
        }
    return fwd_group_launch(args, dev_args, count, reps, share, stream);
}

static int fwd_group_launch(const mpnn_conv_fwd_args *args, const mpnn_conv_fwd_args *dev_args, int count, int reps, int share, void *stream) {
    if (count <= 0) return 0;
    if (!args || !dev_args || count > 4) return MPNN_E_ARG;
    FwdGroupP q = {};
    ConvP hp[4] = {};
    int rows = 0, gxm = 0;
    bool any_idx = false;
    for (int k = 0; k < count; ++k) {
        ConvP &p = hp[k];
        int rc = fill_fwd(&args[k], p);
        if (rc) return rc;
        if ((args[k].idx != nullptr) != (args[0].idx != nullptr)) return MPNN_E_ARG;    // all members routed, or none
        any_idx = any_idx || args[k].idx;
        if (p.n <= 0 || (p.Cout % 16) || p.a.C > 128 || p.Cv > 128 || (p.Cv & 3)) return MPNN_E_SHAPE;
        q.small[k] = p.a.C <= 4;
        if (!q.small[k] && (p.a.C & 3)) return MPNN_E_SHAPE;
        static const int strip_env = [] { const char *e = getenv("MPNN_STRIP"); return e ? atoi(e) : 512; }();     // minimum batch, 0 = off
        // (evaluation batches only: at the training batch a strip per wave leaves the chip half empty -- 14.7 against 13.3 us;
        // co-trained nets count together: `share` nets of n images each fill the chip like one batch of share * n)
        const int kch = (p.a.C >> 4) + (args[k].v ? (p.Cv >> 4) : 0);                  // 16-channel chunks of input
        if (strip_env && (long)p.n * share >= strip_env && q.small[k] && args[k].v && (p.Cv % 16) == 0 && 1 + (p.Cv >> 4) <= MPNN_STRIP_KMAX &&
            p.a.mode == MPNN_ACT_IDENTITY && p.W >= 16 && (p.W % 16) == 0 && (p.H % 4) == 0 && (!args[k].pool_out || !(p.H & 1))) {
            q.gk[k] = 4;  p.n_tiles = conv_grid_x<0>(p.n, p.H, p.W);                    // image + V: the strip body's SMA form
        } else
        if (strip_env && (long)p.n * share >= strip_env && p.W >= 16 && (p.W % 16) == 0 && (p.H % 4) == 0 && (p.a.C % 16) == 0 && p.a.C >= 16 &&
            (!args[k].v || (p.Cv % 16) == 0) && kch <= MPNN_STRIP_KMAX && (!args[k].pool_out || !(p.H & 1))) {
            q.gk[k] = kch == 1 ? 3 : 4;  p.n_tiles = conv_grid_x<0>(p.n, p.H, p.W);        // (64-pixel tiles: the unit of the work shares)
        } else
        if (p.W >= 16 && (p.W % 16) == 0 && (p.H % 4) == 0) { q.gk[k] = 0; p.n_tiles = conv_grid_x<0>(p.n, p.H, p.W); }
        else if (p.W == 8 && p.H == 8) { q.gk[k] = 1; p.n_tiles = conv_grid_x<1>(p.n, 8, 8); }
        else if (p.W == 4 && p.H == 4) { q.gk[k] = 2; p.n_tiles = conv_grid_x<2>(p.n, 4, 4); }
        else return MPNN_E_SHAPE;
        q.gy[k] = p.Cout / 16;
        q.y0[k] = rows;
        rows += q.gy[k];
    }
    // the first conv of a net (image -> 16 channels, no operand V): its own wave-per-tile kernel (conv_first.hip)
    if (count == 1 && reps == 1 && share == 1 && mpnn_first_conv_launch(&args[0], (hipStream_t)stream) == 0) return 0;
    if (count == 1 && (reps > 1 || share > 1) && mpnn_first_conv_launch_rep(&args[0], dev_args, reps, share, (hipStream_t)stream) == 0) return 0;
    // Share the resident workgroup slots between the members in proportion to their work, so that
    // every member is resident from the start.
    const int bytes[5] = {ConvSmem<0, 4, 16>::BYTES, ConvSmem<1, 4, 16>::BYTES, ConvSmem<2, 4, 16>::BYTES, 2048, strip_lds_bytes(MPNN_STRIP_KMAX)};
    int lds = 0;
    for (int k = 0; k < count; ++k) if (bytes[q.gk[k]] > lds) lds = bytes[q.gk[k]];
    // a single deep member on a small map: 128-256 workgroups of 4 waves would leave every SIMD with one
    // wave and nothing to overlap -> K-split body (two thread groups per workgroup, 32-channel units)
    // (read at every launch -- launches are issued once per captured graph --, so a test can switch it per engine)
    const int ks_env = [] { const char *e = getenv("MPNN_FWD_KSPLIT"); return e ? atoi(e) : 1; }();
    // (training launches only: in the evaluation path the body of a conv depends on its shapes and its sample
    // capacity alone, so routed and dense evaluation of a batch agree bit for bit)
    // (one net only: with several nets in the launch there are workgroups enough for every SIMD)
    if (ks_env && reps == 1 && share == 1 && !any_idx && hp[0].a.mode == MPNN_ACT_BN_BATCH && count == 1 && q.gk[0] != 0 && q.gk[0] < 3 && !q.small[0] && (hp[0].a.C % 32) == 0 && (hp[0].Cv % 32) == 0 &&
        hp[0].a.C + hp[0].Cv >= 64) {
        const int gy = q.gy[0];
        int gx = hp[0].n_tiles;
        // (the four-way split: correct, measured NOT faster -- h4 64+64->64 15.4 -> 15.7 us, h4 128->128 15.7 -> 16.6 us in situ:
        // half the unit chain, but sixteen-wave workgroups start later and stage twice the LDS per barrier -- opt-in;
        // read at every launch so that a test can switch it)
        const int ks4_env = [] { const char *e = getenv("MPNN_FWD_KSPLIT4"); return e ? atoi(e) : 0; }();
        const bool ks4 = ks4_env && (hp[0].a.C % 64) == 0 && (hp[0].Cv % 64) == 0 && hp[0].a.C + hp[0].Cv >= 128;
        typedef void (*KsKern)(const mpnn_conv_fwd_args *, const int, const int);
        const KsKern kern = q.gk[0] == 1 ? (ks4 ? fwd_ks_k<1, 4> : fwd_ks_k<1, 2>) : (ks4 ? fwd_ks_k<2, 4> : fwd_ks_k<2, 2>);
        const int lds_ks = q.gk[0] == 1 ? (ks4 ? ConvSmem<1, 4, 16, 4>::BYTES : ConvSmem<1, 4, 16, 2>::BYTES)
                                        : (ks4 ? ConvSmem<2, 4, 16, 4>::BYTES : ConvSmem<2, 4, 16, 2>::BYTES);
        static bool raised = false;
        if (!raised) {                              // (more than the default 64 KB of dynamic LDS)
            (void)hipFuncSetAttribute((const void *)fwd_ks_k<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, ConvSmem<1, 4, 16, 4>::BYTES);
            (void)hipFuncSetAttribute((const void *)fwd_ks_k<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, ConvSmem<2, 4, 16, 4>::BYTES);
            (void)hipFuncSetAttribute((const void *)fwd_ks_k<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, ConvSmem<1, 4, 16, 2>::BYTES);
            (void)hipFuncSetAttribute((const void *)fwd_ks_k<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, ConvSmem<2, 4, 16, 2>::BYTES);
            raised = true;
        }
        const int threads = ks4 ? 1024 : 512;
        const long slots = resident_slots((const void *)kern, lds_ks, threads);
        if ((long)gx * gy > slots) gx = (int)(slots / gy > 0 ? slots / gy : 1);
        gx = xcd_round(gx);
        hipLaunchKernelGGL(kern, dim3(gx * gy), dim3(threads), lds_ks, (hipStream_t)stream, dev_args, gx, xcd_env());
        MPNN_LAUNCH_CHECK();
        return 0;
    }
    bool any_small = false;
    for (int k = 0; k < count; ++k) any_small = any_small || q.small[k];
    // 32-channel output tiles: every member an 8x8 / 4x4 conv with Cout % 32 == 0, evaluation mode, capacity >= MPNN_FWD_WIDE
    static const int wide_env = [] { const char *e = getenv("MPNN_FWD_WIDE"); return e ? atoi(e) : 1024; }();     // 0 = off
    static const int wide_train = [] { const char *e = getenv("MPNN_FWD_WIDE_TRAIN"); return e ? atoi(e) : 0; }();     // experiment: also in multi-member training levels
    bool wide = wide_env > 0;
    for (int k = 0; k < count; ++k)
        wide = wide && (q.gk[k] == 1 || q.gk[k] == 2) && !q.small[k] && (hp[k].Cout % 32) == 0 && (long)hp[k].n * share >= wide_env &&
               (hp[k].a.mode != MPNN_ACT_BN_BATCH || (share > 1 && (count == 1 || wide_train)));
    // (training launches of a co-trained group: a deep 4x4 / 8x8 conv alone in its level, measured at 8 nets x 128 images:
    // h4 64+64->64 38.0 -> 33.7 us, 64->128 37.7 -> 35.2, 128->128 63.5 -> 58.1; in the two-member levels the 8x8 member
    // got slower -- 92 -> 108 us -- and they keep the 16-channel tile)
    if (wide) {
        const int wb[3] = {0, ConvSmem<1, 2, 32>::BYTES, ConvSmem<2, 2, 32>::BYTES};
        lds = 0;
        for (int k = 0; k < count; ++k) { q.gy[k] = hp[k].Cout / 32; if (wb[q.gk[k]] > lds) lds = wb[q.gk[k]]; }
    }
    void (*kern)(const mpnn_conv_fwd_args *, const FwdGroupP) =
        wide ? (any_idx ? fwd_group_k<false, true, true> : fwd_group_k<false, false, true>) :
        any_idx ? (any_small ? fwd_group_k<true, true> : fwd_group_k<false, true>)
                : (any_small ? fwd_group_k<true, false> : fwd_group_k<false, false>);
    // (nets that share the device: no XCD-aware tile order -- its grids are rounded down to multiples of 8 workgroups, a large
    // fraction of slots / share, and the copies' workgroup offsets break the image -> XCD rule anyway: 2 057 -> 2 010 us per
    // joint step of 8 nets with it off)
    const bool use_xcd = reps == 1 && share == 1;
    long slots = resident_slots((const void *)kern, lds) / share;
    if (slots < 1) slots = 1;
    // work of a member = tile-rows x units per tile (16-channel chunks of both operands)
    long total = 0, work[4];
    for (int k = 0; k < count; ++k) {
        const long units = ((hp[k].a.C + 15) >> 4) + (hp[k].v ? ((hp[k].Cv + 15) >> 4) : 0);
        work[k] = (long)hp[k].n_tiles * q.gy[k] * units;
        total += work[k];
    }
    for (int k = 0; k < count; ++k) {
        long g = (slots * work[k]) / (total > 0 ? total : 1) / q.gy[k];      // workgroups per tile-row
        if (g < 1) g = 1;
        if (g > hp[k].n_tiles) g = hp[k].n_tiles;
        if (q.gk[k] >= 3 && g > (hp[k].n_tiles + 3) / 4) g = (hp[k].n_tiles + 3) / 4;       // a wave per strip of >= 4 rows
        q.gx[k] = use_xcd ? xcd_round((int)g) : (int)g;
        if (q.gk[k] >= 3) {
            // rows per strip: the longest (least halo) that still gives every wave of the member's workgroups a strip
            const long cols = (long)hp[k].n * (hp[k].W >> 4);
            int rh = 4;
            for (int r = hp[k].H; r > 4; r >>= 1)
                if (!(r & 1) && hp[k].H % r == 0 && cols * (hp[k].H / r) >= 4L * q.gx[k]) { rh = r; break; }
            { const char *e = getenv("MPNN_STRIP_RH"); if (e && atoi(e) >= 4 && hp[k].H % atoi(e) == 0) rh = atoi(e); }   // (experiments)
            q.rh[k] = rh;
        }
        if (q.gx[k] > gxm) gxm = q.gx[k];
    }
    q.n = count;
    q.xcd = use_xcd ? xcd_env() : 0;
    int n_wg = 0;
    for (int k = 0; k < count; ++k) { q.w0[k] = n_wg; n_wg += q.gx[k] * q.gy[k]; }
    (void)gxm; (void)rows;
    q.reps = reps;  q.wpr = n_wg;
    hipLaunchKernelGGL(kern, dim3(n_wg * reps), dim3(256), lds, (hipStream_t)stream, dev_args, q);
    MPNN_LAUNCH_CHECK();
    return 0;
}
