// Forward conv of a 16-channel map into 16 k channels (a workgroup row per 16 of them) on the big maps (W % 16 == 0: the 32x32 / 16x16 scales of the
// first blocks), no operand V: a WAVE walks a strip of 16 pixels x `rh` rows top to bottom and nothing is shared
// between waves -- no LDS tiles, no barrier.
//
//   * one HALO ROW at a time: lane (li, g) loads the float4 of channels 4g .. 4g+3 of pixel li + dx - 1, dx = 0, 1, 2 --
//     three fully coalesced 1 KB loads per row -- applies BatchNorm + ReLU to them, and the row feeds the MFMAs of
//     the three output rows it belongs to (dy = 0 of row h, dy = 1 of row h-1, dy = 2 of row h-2): 36 MFMAs per
//     halo row, three accumulators rotating.  The loads of the next row are in flight meanwhile;
//   * the 36 weight fragments (9 taps x 4 k-steps) live in registers for the whole kernel;
//   * operands swapped (D = W^T X) as in conv_first.hip: a lane ends with four consecutive channels of a pixel --
//     float4 stores, the 2x2 max-pool a register max over two rows and a DPP lane swap;
//   * same contraction order as the general body (taps ascending, the four k-steps of a tap in order; bias added
//     last) and the same BatchNorm expression: the outputs are bit-identical to mpnn_msconv_fwd's.
// In the general body this layer has one 16-channel unit per 64-pixel tile -- halo staging, two barriers and an
// epilogue per 36 MFMAs of a wave: 58 TFLOP/s at 4 096 images, the largest launch of the evaluation pass.
#pragma once
#include "conv_kernel.h"

struct StripSeq { int ij, ys, xs; };      // image index in the wave's sequence, row segment, column strip

template <bool IDX>
__device__ __forceinline__ void strip16_body(const mpnn_conv_fwd_args &a, const int bx, const int by, const int gx, const int rh,
                                             const int xcd, char *smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int H = a.H, W = a.W, Co = a.Cout, co0 = by * 16;      // this workgroup's 16 output channels
    int n_img = a.n;
    if (a.cnt) { const int c = *a.cnt; n_img = c < a.n ? c : a.n; }        // device-side count of the routed sub-batch
    const int xs_n = W >> 4, ys_n = H / rh, tpi = xs_n * ys_n;

    // BatchNorm + ReLU coefficients of this lane's four channels (the general body's table: mean, gamma * rstd, beta)
    float *cS = (float *)smem;                          // [16][3]
    const int mode = a.a.mode;
    if (mode != MPNN_ACT_IDENTITY) {
        if (tid < 16) {
            const BnC k = bn_coef(a.a, tid);
            cS[tid * 3] = k.m; cS[tid * 3 + 1] = k.gamma * k.rstd; cS[tid * 3 + 2] = k.beta;
        }
        __syncthreads();
    }
    float cm[4], cs[4], cb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        cm[j] = mode != MPNN_ACT_IDENTITY ? cS[(4 * g + j) * 3] : 0.f;
        cs[j] = mode != MPNN_ACT_IDENTITY ? cS[(4 * g + j) * 3 + 1] : 1.f;
        cb[j] = mode != MPNN_ACT_IDENTITY ? cS[(4 * g + j) * 3 + 2] : 0.f;
    }
    // weights: fragment of (tap t, k-step j) = W[t][ci = 4g + j][co = li]; forward pack [tap][ci / 4][Cout][4]
    f32x4 wr[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[t] = *(const f32x4 *)(a.wa_pack + t * 16 * Co + (g * Co + co0 + li) * 4);
    const f32x4 bias4 = *(const f32x4 *)(a.bias + co0 + g * 4);

    // strips of this wave: jw, jw + nw, ... of the launch's sequence (XCD-aware: of its own XCD's images, conv_kernel.h)
    const bool xa = xcd != 0 && (a.n & 31) == 0 && (gx & 7) == 0 && !IDX;
    const int xcd_id = blockIdx.x & 7;
    const int jw = xa ? (bx >> 3) * 4 + wid : bx * 4 + wid;
    const int nw = xa ? (gx >> 3) * 4 : gx * 4;
    const int jn = (xa ? (a.n >> 3) : n_img) * tpi;
    auto image = [&](int ij) { return xa ? ((ij >> 2) << 5) + 4 * xcd_id + (ij & 3) : ij; };

    const long row_b = (long)W * 64;                       // bytes between two rows of a 16-channel map
    const int off1 = (li * 16 + g * 4) * 4;                // this lane's float4 of pixel x0 + li
    [[maybe_unused]] float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const bool stats = a.out_sum != nullptr, pool = a.pool_out != nullptr;

    for (int j = jw; j < jn; j += nw) {
        const int ij = j / tpi, rem = j - ij * tpi, ys = rem / xs_n, xs = rem - ys * xs_n;
        const int slot = image(ij);
        const int n = IDX ? a.idx[slot] : slot;
        const int y0 = ys * rh, x0 = xs * 16;
        const bool left = x0 == 0, right = x0 + 16 == W;
        const bool zl = left && li == 0, zr = right && li == 15;
        // lane offsets of the three column shifts (edge lanes of edge strips read the middle one and are zeroed)
        const int o0 = off1 - 64 + (64 & -(int)zl), o2 = off1 + 64 - (64 & -(int)zr);
        const char *xin = (const char *)a.a.x + (((long)n * H) * W + x0) * 64;
        float *outp = a.out + (((long)n * H) * W + x0 + li) * Co + co0 + g * 4;
        float *poolp = pool ? a.pool_out + (((long)n * (H >> 1)) * (W >> 1) + (x0 >> 1) + (li >> 1)) * Co + co0 + g * 4 : nullptr;

        // halo row y: raw loads (a row outside the image reads row 0 and is zeroed on use)
        auto load_row = [&](int y, f32x4 *r) {
            const char *rp = xin + ((unsigned)y < (unsigned)H ? y : 0) * row_b;
            r[0] = *(const f32x4 *)(rp + o0);
            r[1] = *(const f32x4 *)(rp + off1);
            r[2] = *(const f32x4 *)(rp + o2);
        };
        // BatchNorm + ReLU (the general body's expression), zero padding AFTER it
        auto prep_row = [&](int y, f32x4 *r) {
            const bool rz = (unsigned)y >= (unsigned)H;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const bool z = rz || (dx == 0 && zl) || (dx == 2 && zr);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float v = r[dx][c];
                    if (mode != MPNN_ACT_IDENTITY) v = fmaxf((v - cm[c]) * cs[c] + cb[c], 0.f);
                    r[dx][c] = z ? 0.f : v;
                }
            }
        };
        // the MFMAs of one prepared halo row: tap row 2 of `up`, 1 of `mid`, 0 of `dn` -- INTERLEAVED, so that consecutive
        // MFMAs never wait for each other's result (every accumulator still sees its own taps in ascending order)
        auto mac3 = [&](f32x4 &up, f32x4 &mid, f32x4 &dn, const f32x4 *r, bool on_up, bool on_mid, bool on_dn) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (on_up) up = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[6 + dx][c], r[dx][c], up, 0, 0, 0);
                    if (on_mid) mid = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[3 + dx][c], r[dx][c], mid, 0, 0, 0);
                    if (on_dn) dn = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[dx][c], r[dx][c], dn, 0, 0, 0);
                }
        };
        f32x4 prev_out = {0.f, 0.f, 0.f, 0.f};
        // output row y is complete: bias, store, statistics, pooled row every second row
        auto finish = [&](f32x4 acc, int y) {
            acc += bias4;
            *(f32x4 *)(outp + (long)y * W * Co) = acc;
            if (stats) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { s1[c] += acc[c]; s2[c] += acc[c] * acc[c]; }
            }
            if (pool) {
                if (y & 1) {
                    f32x4 m;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float q = fmaxf(prev_out[c], acc[c]);
                        const float o = __builtin_bit_cast(float, dpp_i<MPNN_DPP_QUAD_XOR1>(__builtin_bit_cast(int, q)));
                        m[c] = fmaxf(q, o);
                    }
                    *(f32x4 *)(poolp + (long)(y >> 1) * (W >> 1) * Co) = m;      // (both lanes of a pair store the same value)
                }
                prev_out = acc;
            }
        };

        // halo rows y0 - 1 .. y0 + rh; row k+1 is in flight while row k is multiplied (MPNN_STRIP_AHEAD = 2, two rows
        // ahead in a third register set: 300 against 259 us at 4 096 images in alternating runs on one box -- the 12
        // registers spill inside the group kernel's 128-register budget)
#ifndef MPNN_STRIP_AHEAD
#define MPNN_STRIP_AHEAD 1
#endif
        f32x4 ra[3], rb[3];
        [[maybe_unused]] f32x4 rc[3];
        f32x4 A0 = {0.f, 0.f, 0.f, 0.f}, A1 = A0, A2 = A0;      // accumulators of output rows k, k-1, k-2 (relative to halo row k)
        load_row(y0 - 1, ra);
        if (MPNN_STRIP_AHEAD == 2) load_row(y0, rb);
        // one step: halo row k = y0 - 1 + i in `cur`; row k + AHEAD is requested into `nxt` (the set that is free)
        auto step = [&](int i, f32x4 *cur, f32x4 *nxt, f32x4 &Adn, f32x4 &Amid, f32x4 &Aup) {
            // Adn: output row i (this halo row is its dy = 0), Amid: row i - 1 (dy = 1), Aup: row i - 2 (dy = 2)
            const int y = y0 - 1 + i;
            if (i + MPNN_STRIP_AHEAD <= rh + 1) load_row(y + MPNN_STRIP_AHEAD, nxt);
            prep_row(y, cur);
            const bool on_up = i >= 2, on_mid = i >= 1 && i <= rh, on_dn = i <= rh - 1;          // (uniform)
            if (on_dn) Adn = f32x4{0.f, 0.f, 0.f, 0.f};
            if (on_up && on_mid && on_dn) mac3(Aup, Amid, Adn, cur, true, true, true);             // (the interior of the strip)
            else mac3(Aup, Amid, Adn, cur, on_up, on_mid, on_dn);
            if (on_up) { mfma_drain(); finish(Aup, y0 + i - 2); }
        };
#if MPNN_STRIP_AHEAD == 2
        for (int i = 0; i <= rh + 1; i += 3) {
            step(i, ra, rc, A0, A2, A1);
            if (i + 1 <= rh + 1) step(i + 1, rb, ra, A1, A0, A2);
            if (i + 2 <= rh + 1) step(i + 2, rc, rb, A2, A1, A0);
        }
#else
        for (int i = 0; i <= rh + 1; i += 6) {                  // (period 6: two register sets x three accumulators)
            step(i, ra, rb, A0, A2, A1);
            if (i + 1 <= rh + 1) step(i + 1, rb, ra, A1, A0, A2);
            if (i + 2 <= rh + 1) step(i + 2, ra, rb, A2, A1, A0);
            if (i + 3 <= rh + 1) step(i + 3, rb, ra, A0, A2, A1);
            if (i + 4 <= rh + 1) step(i + 4, ra, rb, A1, A0, A2);
            if (i + 5 <= rh + 1) step(i + 5, rb, ra, A2, A1, A0);
        }
#endif
    }

    if (a.out_sum) {
        // per channel: over the 16 pixel lanes of the row, then over the four waves (LDS), one fp64 atomic per channel
        // and workgroup into the workgroup's slot -- as the general body does
        double *red = (double *)(smem + 256);              // [4][16][2]
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double a1 = (double)s1[c], a2 = (double)s2[c];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { a1 += __shfl_xor(a1, m); a2 += __shfl_xor(a2, m); }
            if (li == 0) { red[(wid * 16 + g * 4 + c) * 2] = a1; red[(wid * 16 + g * 4 + c) * 2 + 1] = a2; }
        }
        __syncthreads();
        if (tid < 16) {
            double a1 = 0.0, a2 = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { a1 += red[(w * 16 + tid) * 2]; a2 += red[(w * 16 + tid) * 2 + 1]; }
            const int nslot = a.out_nslot < 1 ? 1 : (a.out_nslot > MPNN_BN_SLOTS ? MPNN_BN_SLOTS : a.out_nslot);
            double *slot = a.out_sum + (size_t)(bx % nslot) * 2 * Co;
            atomicAdd(slot + co0 + tid, a1);
            atomicAdd(slot + Co + co0 + tid, a2);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same walk for K = 2 .. 3 sixteen-channel chunks of input (32-channel operand A, or 16 / 32 channels of A plus
// the pooled finer map V): a halo row is processed chunk by chunk into the same three accumulators.  The weight
// fragments no longer fit the registers (9 K float4 per lane): the workgroup keeps them in LDS ([chunk][tap][lane],
// conflict-free 16-byte reads, one per four MFMAs), as the BatchNorm coefficients of A's channels.
// An output row sums its taps in the order (dy, chunk, dx, k-step); the general body sums (chunk, dy, dx, k-step):
// the results agree to fp32 summation order, not bit for bit -- which is why the choice between the bodies depends
// only on the launch's sample CAPACITY: dense and routed evaluation of a batch take the same body and stay identical.
// ---------------------------------------------------------------------------------------------------------------
#define MPNN_STRIP_KMAX 3
static inline int strip_lds_bytes(int K) { return 2048 + 1024 + K * 9 * 64 * 16; }      // red + coefficients + weights

// SMA: operand A is the 1..3-channel pyramid image of block 0 (ToPyramid's strided pick, no BatchNorm): chunk 0 is ONE
// k-step per tap (lane (li, g) supplies channel g), the pooled finer map V follows as usual.
template <bool IDX, bool SMA = false>
__device__ __forceinline__ void stripk_body(const mpnn_conv_fwd_args &a, const int bx, const int by, const int gx, const int rh,
                                            const int xcd, char *smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int H = a.H, W = a.W, Co = a.Cout, co0 = by * 16;
    const int Ca = a.a.C, Cv = a.v ? a.Cv : 0, KA = SMA ? 1 : Ca >> 4, KV = Cv >> 4, K = KA + KV;
    const int sh = SMA ? a.a.shift : 0;
    int n_img = a.n;
    if (a.cnt) { const int c = *a.cnt; n_img = c < a.n ? c : a.n; }
    const int xs_n = W >> 4, ys_n = H / rh, tpi = xs_n * ys_n;

    double *red = (double *)smem;                          // [4][16][2]   (statistics, at the end)
    float *cS = (float *)(smem + 2048);                     // [3][Ca]: mean, gamma * rstd, beta of operand A's channels
    f32x4 *wl = (f32x4 *)(smem + 2048 + 1024);              // [K][9][64]
    const int mode = a.a.mode;
    if (!SMA && mode != MPNN_ACT_IDENTITY && tid < Ca) {
        const BnC k = bn_coef(a.a, tid);
        cS[tid] = k.m; cS[Ca + tid] = k.gamma * k.rstd; cS[2 * Ca + tid] = k.beta;
    }
    // fragment (chunk kq, tap t) of lane (li = co, g): W[t][ci = 16 kq + 4g + j][co], j = 0..3: one float4 of the pack
    for (int f = tid; f < K * 9 * 64; f += 256) {
        const int l = f & 63, ft = f >> 6, kq = ft / 9, t = ft - kq * 9, fl = l & 15, fg = l >> 4;
        const bool isv = kq >= KA;
        const float *pk = isv ? a.wv_pack : a.wa_pack;
        const int nch = isv ? KV : KA, kk = isv ? kq - KA : kq;
        if (SMA && !isv) {                                  // the image: W[t][ci = fg][co] in component 0
            const float w = fg < Ca ? pk[(size_t)t * 16 * Co + (co0 + fl) * 4 + fg] : 0.f;
            wl[f] = f32x4{w, 0.f, 0.f, 0.f};
            continue;
        }
        wl[f] = *(const f32x4 *)(pk + (size_t)t * nch * 16 * Co + ((kk * 4 + fg) * Co + co0 + fl) * 4);
    }
    __syncthreads();
    const f32x4 bias4 = *(const f32x4 *)(a.bias + co0 + g * 4);

    const bool xa = xcd != 0 && (a.n & 31) == 0 && (gx & 7) == 0 && !IDX;
    const int xcd_id = blockIdx.x & 7;
    const int jw = xa ? (bx >> 3) * 4 + wid : bx * 4 + wid;
    const int nw = xa ? (gx >> 3) * 4 : gx * 4;
    const int jn = (xa ? (a.n >> 3) : n_img) * tpi;
    auto image = [&](int ij) { return xa ? ((ij >> 2) << 5) + 4 * xcd_id + (ij & 3) : ij; };
    [[maybe_unused]] float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const bool stats = a.out_sum != nullptr, pool = a.pool_out != nullptr;

    for (int j = jw; j < jn; j += nw) {
        const int ij = j / tpi, rem = j - ij * tpi, ys = rem / xs_n, xs = rem - ys * xs_n;
        const int slot = image(ij);
        const int n = IDX ? a.idx[slot] : slot;
        const int y0 = ys * rh, x0 = xs * 16;
        const bool zl = x0 == 0 && li == 0, zr = x0 + 16 == W && li == 15;
        const int pl = li - 1 + (int)zl, pr = li + 1 - (int)zr;            // pixel of the dx = 0 / 2 operand (edge lanes: the middle one)
        const long img_px = ((long)n * H) * W + x0;
        float *outp = a.out + (img_px + li) * Co + co0 + g * 4;
        float *poolp = pool ? a.pool_out + (((long)n * (H >> 1)) * (W >> 1) + (x0 >> 1) + (li >> 1)) * Co + co0 + g * 4 : nullptr;

        // unit (halo row y, chunk kq): raw loads; a row outside the image reads row 0 and is zeroed on use
        auto load_unit = [&](int y, int kq, f32x4 *r) {
            const bool isv = kq >= KA;
            if (SMA && !isv) {                              // (uniform) one channel per lane, strided pick of the pyramid
                const int yy = (unsigned)y < (unsigned)H ? y : 0, gc = g < Ca ? g : Ca - 1;
                const float *rp = a.a.x + ((((long)n * (H << sh) + ((long)yy << sh)) * (W << sh)) + ((long)x0 << sh)) * Ca + gc;
                r[0] = f32x4{rp[(long)(pl << sh) * Ca], 0.f, 0.f, 0.f};
                r[1] = f32x4{rp[(long)(li << sh) * Ca], 0.f, 0.f, 0.f};
                r[2] = f32x4{rp[(long)(pr << sh) * Ca], 0.f, 0.f, 0.f};
                return;
            }
            const float *src = isv ? a.v : a.a.x;
            const int C = isv ? Cv : Ca, ch = (isv ? kq - KA : kq) * 16 + g * 4;
            const float *rp = src + (img_px + (long)((unsigned)y < (unsigned)H ? y : 0) * W) * C + ch;
            r[0] = *(const f32x4 *)(rp + pl * C);
            r[1] = *(const f32x4 *)(rp + li * C);
            r[2] = *(const f32x4 *)(rp + pr * C);
        };
        auto prep_unit = [&](int y, int kq, f32x4 *r) {
            const bool rz = (unsigned)y >= (unsigned)H;
            const bool bn = !SMA && kq < KA && mode != MPNN_ACT_IDENTITY;      // (uniform)
            f32x4 cm = {0.f, 0.f, 0.f, 0.f}, cs = cm, cb = cm;
            if (bn) {
                const int c = kq * 16 + g * 4;
                cm = *(const f32x4 *)(cS + c); cs = *(const f32x4 *)(cS + Ca + c); cb = *(const f32x4 *)(cS + 2 * Ca + c);
            }
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const bool z = rz || (dx == 0 && zl) || (dx == 2 && zr);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float v = r[dx][c];
                    if (bn) v = fmaxf((v - cm[c]) * cs[c] + cb[c], 0.f);
                    r[dx][c] = z ? 0.f : v;
                }
            }
        };
        auto mac3 = [&](f32x4 &up, f32x4 &mid, f32x4 &dn, const f32x4 *r, int kq, bool on_up, bool on_mid, bool on_dn) {
            const f32x4 *wk = wl + kq * 9 * 64 + lane;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const f32x4 w0 = wk[dx * 64], w1 = wk[(3 + dx) * 64], w2 = wk[(6 + dx) * 64];
                auto kstep = [&](int c) {
                    if (on_up) up = __builtin_amdgcn_mfma_f32_16x16x4f32(w2[c], r[dx][c], up, 0, 0, 0);
                    if (on_mid) mid = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[c], r[dx][c], mid, 0, 0, 0);
                    if (on_dn) dn = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[c], r[dx][c], dn, 0, 0, 0);
                };
                kstep(0);
                if (!(SMA && kq == 0)) { kstep(1); kstep(2); kstep(3); }       // (uniform: the image chunk is one k-step)
            }
        };
        f32x4 prev_out = {0.f, 0.f, 0.f, 0.f};
        auto finish = [&](f32x4 acc, int y) {
            acc += bias4;
            *(f32x4 *)(outp + (long)y * W * Co) = acc;
            if (stats) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { s1[c] += acc[c]; s2[c] += acc[c] * acc[c]; }
            }
            if (pool) {
                if (y & 1) {
                    f32x4 m;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float q = fmaxf(prev_out[c], acc[c]);
                        const float o = __builtin_bit_cast(float, dpp_i<MPNN_DPP_QUAD_XOR1>(__builtin_bit_cast(int, q)));
                        m[c] = fmaxf(q, o);
                    }
                    *(f32x4 *)(poolp + (long)(y >> 1) * (W >> 1) * Co) = m;
                }
                prev_out = acc;
            }
        };

        // units in the order (halo row, chunk); unit u + 1 is in flight while unit u is multiplied
        f32x4 cur[3], nxt[3];
        f32x4 A0 = {0.f, 0.f, 0.f, 0.f}, A1 = A0, A2 = A0;
        load_unit(y0 - 1, 0, nxt);
        auto row = [&](int i, f32x4 &Adn, f32x4 &Amid, f32x4 &Aup) {
            const int y = y0 - 1 + i;
            const bool on_up = i >= 2, on_mid = i >= 1 && i <= rh, on_dn = i <= rh - 1;          // (uniform)
            if (on_dn) Adn = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int kq = 0; kq < K; ++kq) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) cur[dx] = nxt[dx];
                const bool last = kq + 1 == K;
                if (!last || i + 1 <= rh + 1) load_unit(last ? y + 1 : y, last ? 0 : kq + 1, nxt);
                prep_unit(y, kq, cur);
                if (on_up && on_mid && on_dn) mac3(Aup, Amid, Adn, cur, kq, true, true, true);
                else mac3(Aup, Amid, Adn, cur, kq, on_up, on_mid, on_dn);
            }
            if (on_up) { mfma_drain(); finish(Aup, y0 + i - 2); }
        };
        for (int i = 0; i <= rh + 1; i += 3) {
            row(i, A0, A2, A1);
            if (i + 1 <= rh + 1) row(i + 1, A1, A0, A2);
            if (i + 2 <= rh + 1) row(i + 2, A2, A1, A0);
        }
    }

    if (a.out_sum) {
        __syncthreads();                                    // (the weights in LDS are not read any more; `red` is its own area anyway)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double a1 = (double)s1[c], a2 = (double)s2[c];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) { a1 += __shfl_xor(a1, m); a2 += __shfl_xor(a2, m); }
            if (li == 0) { red[(wid * 16 + g * 4 + c) * 2] = a1; red[(wid * 16 + g * 4 + c) * 2 + 1] = a2; }
        }
        __syncthreads();
        if (tid < 16) {
            double a1 = 0.0, a2 = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { a1 += red[(w * 16 + tid) * 2]; a2 += red[(w * 16 + tid) * 2 + 1]; }
            const int nslot = a.out_nslot < 1 ? 1 : (a.out_nslot > MPNN_BN_SLOTS ? MPNN_BN_SLOTS : a.out_nslot);
            double *slot = a.out_sum + (size_t)(bx % nslot) * 2 * Co;
            atomicAdd(slot + co0 + tid, a1);
            atomicAdd(slot + Co + co0 + tid, a2);
        }
    }
}
