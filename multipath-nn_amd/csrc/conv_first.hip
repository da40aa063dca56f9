// The first conv of the net (block 0, finest scale): a 1..3-channel image -> 16 channels, no operand V.
//
// In the general body (conv_kernel.h) this layer is all overhead: 9 MFMAs per wave and tile under ~400 vector
// instructions of staging, LDS traffic, barriers and a scalar epilogue -- issue-bound at 15 TFLOP/s, 243 us for
// 4 096 images where its memory traffic (input 50 MB, output 268 MB, pooled output 67 MB) is worth ~85 us.
// Here a WAVE owns a 4 x 16 pixel tile and nothing is shared between waves:
//   * no LDS memory, no barrier: a halo row of the tile (18 pixels x C floats, contiguous) is ONE coalesced dword
//     load per lane, six per tile; the operand of M-tile row r and tap (dy, dx) is halo row r + dy shifted by dx
//     pixels -- a lane permutation (ds_bpermute) of that register, 18 per tile.  (Loading the 18 operands directly,
//     one 12-byte-stride dword load each, was bound by the address coalescer: 99 us for the loads alone at 4 096
//     images.)  The next tile's six loads are in flight under the MFMAs of the current one;
//   * the 9 weight fragments live in registers for the whole kernel;
//   * operands swapped (D = W^T X): a lane ends up with FOUR CONSECUTIVE CHANNELS of one pixel -- float4 stores
//     (1 KB contiguous per instruction), the 2x2 max-pool is a register max over two rows and one DPP lane swap;
//   * same contraction order as the general body (tap by tap, bias added last): the outputs are bit-identical to
//     mpnn_msconv_fwd's; the batch statistics are the same sums in another order.
// Measured at 4 096 images: 243 -> 101 us (the training launch 13.4 -> 11 us).  What is left: the stores alone run at
// the rate of a fill kernel (54 us), but the MFMAs (+26 us) and the loads (+23 us) ADD to that instead of hiding under
// it -- independent of the occupancy (2 .. 8 workgroups per CU), of staggered starts, of non-temporal hints, of an
// LDS round trip instead of the lane permutes and of the prefetch distance: not understood.
#include "conv_kernel.h"

struct FirstP {
    const float *x, *w, *bias;
    float *out, *pool_out;
    double *out_sum;
    int n, H, W, C, nslot, n_tiles, xcd;
};

struct FirstTile { int n0, y0, x0; };

// REP (co-trained nets, mpnn_msconv_fwd_group_rep): the grid is copies of `wpr` workgroups, copy r runs the first conv of
// net r -- its record tab[r], the geometry of p0.
template <bool STATS, bool POOL, bool REP = false>
__global__ __launch_bounds__(256) void fwd_first_k(const FirstP p0, const mpnn_conv_fwd_args *__restrict__ tab = nullptr, const int wpr = 0) {
    FirstP p = p0;
    int bid = blockIdx.x, gdim = gridDim.x;
    if constexpr (REP) {
        const int rep = bid / wpr;
        bid -= rep * wpr;  gdim = wpr;
        const mpnn_conv_fwd_args *a = tab + rep;
        p.x = a->a.x; p.w = a->wa_pack; p.bias = a->bias; p.out = a->out; p.pool_out = a->pool_out; p.out_sum = a->out_sum;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int C = p.C, H = p.H, W = p.W;
    const int gc = g < C ? g : C - 1;                     // lanes of the padding channels read channel C-1 (times a zero weight)
    const int tx_n = W >> 4, tpi = tx_n * (H >> 2);
    const long row_b = (long)W * C * 4;                    // bytes between two rows of the image

    // the weights: fragment of tap t = W[t][ci = g][co = li] (forward pack: [tap][16 ci / 4][Cout][4])
    float wr[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) { const float v = p.w[t * 256 + li * 4 + gc]; wr[t] = g < C ? v : 0.f; }
    const f32x4 bias4 = *(const f32x4 *)(p.bias + g * 4);

    // A halo row of a tile is 18 pixels x C floats, contiguous in memory: lane e holds float e of it (C <= 3: 54
    // floats); the operand of column shift dx is then a lane permutation of that register -- float (li + dx) C + g.
    const int ne = 18 * C;
    const int le = lane < ne ? lane : ne - 1;
    const int o_mid = (le - C) * 4;                                            // byte offset from pixel x0 of the row
    const int o_left = (le < C ? 0 : le - C) * 4;                              // x0 == 0: pixel -1 does not exist
    const int o_right = (le >= 17 * C ? 17 * C - 1 - C : le - C) * 4;          // x0 + 16 == W: neither does pixel W
    int perm[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) perm[dx] = ((li + dx) * C + gc) * 4;

    // Tiles of this wave: jw, jw + nw, jw + 2 nw, ... of the launch's tile sequence (XCD-aware launches: of the sequence
    // of their own XCD, image 32 q + 4 xcd + i as in conv_kernel.h's xcd_tile) -- at any moment the resident waves
    // write one compact window of the output, which the memory side likes better than a private stream per wave
    // (123 against 132 us at 4 096 images).  The walk is incremental (column, row, image carries): no division
    // in the loop.
    const bool xa = p.xcd != 0 && (p.n & 31) == 0 && (gdim & 7) == 0;
    const int xcd_id = bid & 7;
    const int jw = xa ? (int)(bid >> 3) * 4 + wid : (int)bid * 4 + wid;
    const int nw = xa ? (int)(gdim >> 3) * 4 : (int)gdim * 4;
    const int jn = xa ? (p.n_tiles >> 3) : p.n_tiles;
    const int c0 = jw, c1 = jn, cd = nw;
    const int ty_n = H >> 2;
    const int d_img = nw / tpi, d_rem = nw - d_img * tpi, d_ty = d_rem / tx_n, d_tx = d_rem - d_ty * tx_n;
    auto image = [&](int ij) { return xa ? ((ij >> 2) << 5) + 4 * xcd_id + (ij & 3) : ij; };
    auto origin = [&](int j, int &ij) {
        FirstTile r;
        ij = j / tpi;
        const int rem = j - ij * tpi, ty = rem / tx_n;
        r.n0 = image(ij); r.y0 = ty * 4; r.x0 = (rem - ty * tx_n) * 16;
        return r;
    };
    auto advance = [&](FirstTile &r, int &ij) {             // + nw tiles
        int tx = (r.x0 >> 4) + d_tx, ty = (r.y0 >> 2) + d_ty;
        if (tx >= tx_n) { tx -= tx_n; ++ty; }
        ij += d_img;
        if (ty >= ty_n) { ty -= ty_n; ++ij; }
        r.x0 = tx << 4; r.y0 = ty << 2; r.n0 = image(ij);
    };
    // 6 raw loads of a tile; rows / pixels outside the image read a neighbour inside it and are zeroed on use
    auto load = [&](const FirstTile &q, float *rw) {
        const bool left = q.x0 == 0, right = q.x0 + 16 == W;
        // (mask arithmetic, not `?:`: hipcc turns a select between lane-varying values on uniform conditions into a
        // table in memory and a flat load per use)
        const int o = o_mid + ((o_left - o_mid) & -(int)left) + ((o_right - o_mid) & -(int)right);
        const char *base = (const char *)p.x + (((long)q.n0 * H + q.y0) * W + q.x0) * C * 4;
#pragma unroll
        for (int hr = 0; hr < 6; ++hr) {
            int yr = hr - 1;
            if (hr == 0 && q.y0 == 0) yr = 0;                   // (uniform)
            if (hr == 5 && q.y0 + 4 == H) yr = 3;
            rw[hr] = *(const float *)(base + yr * row_b + o);
        }
    };

    [[maybe_unused]] float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    // registers -> operands: rows / pixels outside the image become zeros, then 18 lane permutations
    auto take = [&](const FirstTile &q, const float *rw, float (*v)[3]) {
        const bool top = q.y0 == 0, bot = q.y0 + 4 == H;
        const bool edge = (q.x0 == 0 && lane < C) || (q.x0 + 16 == W && lane >= 17 * C);
#pragma unroll
        for (int hr = 0; hr < 6; ++hr) {
            const bool z = edge || (hr == 0 && top) || (hr == 5 && bot);
            const int r = __builtin_bit_cast(int, z ? 0.f : rw[hr]);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) v[hr][dx] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(perm[dx], r));
        }
    };
    // MFMAs, stores, statistics and pooled map of one tile
    auto compute = [&](const FirstTile &q, const float (*v)[3]) {
        f32x4 acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3, dx = t - dy * 3;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[t], v[r + dy][dx], acc[r], 0, 0, 0);
        }
        mfma_drain();
        // D rows = channels: acc[r][k] = channel 4 g + k of pixel (y0 + r, x0 + li)
        float *o = p.out + ((((long)q.n0 * H + q.y0) * W + q.x0) * 16) + (li * 16 + g * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[r] += bias4;
            *(f32x4 *)(o + (long)r * W * 16) = acc[r];
            if constexpr (STATS) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { s1[k] += acc[r][k]; s2[k] += acc[r][k] * acc[r][k]; }
            }
        }
        if constexpr (POOL) {
            const int H2 = H >> 1, W2 = W >> 1;
            float *po = p.pool_out + ((((long)q.n0 * H2 + (q.y0 >> 1)) * W2 + (q.x0 >> 1)) * 16) + ((li >> 1) * 16 + g * 4);
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                f32x4 m;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float a = fmaxf(acc[2 * rr][k], acc[2 * rr + 1][k]);
                    const float b = __builtin_bit_cast(float, dpp_i<MPNN_DPP_QUAD_XOR1>(__builtin_bit_cast(int, a)));
                    m[k] = fmaxf(a, b);
                }
                // (both lanes of a pair hold the same maximum and store it to the same place: no branch around the
                // store, so the waits for later tiles' loads can be counted past these stores)
                *(f32x4 *)(po + (long)rr * W2 * 16) = m;
            }
        }
    };

    // Software pipeline, loads two tiles ahead in two register sets: an iteration is [MFMAs and stores of tile t |
    // tile t+1 moves from its load registers to the operand registers | tile t+3 is requested into the set just
    // freed].  Vector-memory operations complete IN ORDER on gfx9 (one vmcnt for loads and stores), so waiting for
    // a load also waits for every store issued before it: with the loads of tile t+1 issued two iterations ahead
    // the move waits with vmcnt(18) -- behind the stores of tile t-2, not of tile t-1.  The first iteration is peeled
    // so that every way into the loop's wait has the same 18 operations behind the loads it needs (where paths
    // with different counts merge the compiler takes the smallest: with the wait at the top of an un-peeled loop it
    // was vmcnt(0) and every tile waited for the previous tile's stores to reach memory).
    const int nt = c0 < c1 ? (c1 - 1 - c0) / cd + 1 : 0;
    if (nt > 0) {
        int ij = 0, made = 1;
        FirstTile gq = origin(c0, ij);
        auto next = [&]() { if (made < nt) advance(gq, ij); ++made; return gq; };     // past the end: the last tile again
        float ra[6], rb[6], v[6][3];
        FirstTile qa = gq;
        load(qa, ra);
        take(qa, ra, v);
        FirstTile qb = next();  load(qb, rb);
        FirstTile qc = next();  load(qc, ra);
        compute(qa, v);  take(qb, rb, v);  qa = qb;  qb = qc;  qc = next();  load(qc, rb);
        __builtin_amdgcn_sched_barrier(0);
        int i = 1;
        for (; i + 1 < nt; i += 2) {       // (two tiles per trip, the odd one after the loop: no short way back to a wait)
            compute(qa, v);  take(qb, ra, v);  qa = qb;  qb = qc;  qc = next();  load(qc, ra);
            __builtin_amdgcn_sched_barrier(0);
            compute(qa, v);  take(qb, rb, v);  qa = qb;  qb = qc;  qc = next();  load(qc, rb);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (i < nt) compute(qa, v);
    }

    if constexpr (STATS) {
        // per channel: over the 16 pixel lanes of the row (butterfly), then over the four waves (LDS), one fp64
        // atomic per channel and workgroup into the workgroup's slot -- as the general body does
        __shared__ double red[4][16][2];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double a1 = (double)s1[k], a2 = (double)s2[k];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { a1 += __shfl_xor(a1, m); a2 += __shfl_xor(a2, m); }
            if (li == 0) { red[wid][g * 4 + k][0] = a1; red[wid][g * 4 + k][1] = a2; }
        }
        __syncthreads();
        if (tid < 16 && p.out_sum) {
            double a1 = 0.0, a2 = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { a1 += red[w][tid][0]; a2 += red[w][tid][1]; }
            double *slot = p.out_sum + (size_t)(bid % p.nslot) * 2 * 16;
            atomicAdd(slot + tid, a1);
            atomicAdd(slot + 16 + tid, a2);
        }
    }
}

static int first_conv_launch(const mpnn_conv_fwd_args *a, const mpnn_conv_fwd_args *dev_args, int reps, int share, hipStream_t st);

// Takes the launch if the record is the first conv of a net (see the top of the file); 0 = launched, 1 = not mine.
int mpnn_first_conv_launch(const mpnn_conv_fwd_args *a, hipStream_t st) { return first_conv_launch(a, nullptr, 1, 1, st); }
// ... of `reps` nets at once: a = net 0's record (the others have its shapes: checked by the caller), dev_args = the
// reps records in device memory; every net's grid is sized for resident slots / share.
int mpnn_first_conv_launch_rep(const mpnn_conv_fwd_args *a, const mpnn_conv_fwd_args *dev_args, int reps, int share, hipStream_t st) {
    return first_conv_launch(a, dev_args, reps, share, st);
}

static int first_conv_launch(const mpnn_conv_fwd_args *a, const mpnn_conv_fwd_args *dev_args, int reps, int share, hipStream_t st) {
    static const int on = [] { const char *e = getenv("MPNN_FIRST_CONV"); return e ? atoi(e) : 1; }();
    if (!on || a->idx || a->cnt || a->v || a->Cout != 16 || a->a.C < 1 || a->a.C > 3 || a->a.shift != 0 ||
        a->a.mode != MPNN_ACT_IDENTITY) return 1;
    if (a->W < 16 || (a->W % 16) || (a->H % 4) || a->n <= 0) return 1;
    if (a->pool_out && ((a->H & 1) || (a->W & 1))) return 1;
    FirstP p = {};
    p.x = a->a.x; p.w = a->wa_pack; p.bias = a->bias; p.out = a->out; p.pool_out = a->pool_out; p.out_sum = a->out_sum;
    p.n = a->n; p.H = a->H; p.W = a->W; p.C = a->a.C;
    p.nslot = a->out_nslot < 1 ? 1 : (a->out_nslot > MPNN_BN_SLOTS ? MPNN_BN_SLOTS : a->out_nslot);
    p.n_tiles = a->n * (a->W >> 4) * (a->H >> 2);
    const bool use_xcd = reps == 1 && share == 1;       // (see fwd_group_launch)
    p.xcd = use_xcd ? xcd_env() : 0;
    const bool stats = a->out_sum != nullptr, pool = a->pool_out != nullptr;
    typedef void (*FirstKern)(const FirstP, const mpnn_conv_fwd_args *, const int);
    FirstKern kern = reps > 1 ? (stats ? (pool ? fwd_first_k<true, true, true> : fwd_first_k<true, false, true>)
                                       : (pool ? fwd_first_k<false, true, true> : fwd_first_k<false, false, true>))
                              : (stats ? (pool ? fwd_first_k<true, true> : fwd_first_k<true, false>)
                                       : (pool ? fwd_first_k<false, true> : fwd_first_k<false, false>));
    long wgs = resident_slots((const void *)kern, 0) / share;
    const long need = (p.n_tiles + 3) / 4;
    if (wgs > need) wgs = need;
    if (wgs < 1) wgs = 1;
    const int grid = use_xcd ? xcd_round((int)wgs) : (int)wgs;
    hipLaunchKernelGGL(kern, dim3(grid * reps), dim3(256), 0, st, p, dev_args, grid);
    MPNN_LAUNCH_CHECK();
    return 0;
}
