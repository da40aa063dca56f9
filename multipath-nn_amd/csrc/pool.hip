// mpnn_maxpool_fwd / mpnn_maxpool_bwd: the single-scale MaxPool and GlobalMaxPool layers (reference
// scripts/lib/layer_types.py:86-100) on pre-activation NHWC maps.  Unused by every shipped spec; here so that a Chain
// of Conv / Rect / MaxPool / GlobalMaxPool / LinTrans (lib/_plan_conv.py) runs instead of being refused.
//
//   MaxPool        tf.nn.max_pool(x, ksize, strides, 'SAME').  NOTE: the reference passes its hypers in the order
//                  (strides, k_shape) -- so the WINDOW is `stride` and the STEP is `supp` (layer_types.py:90-94); the
//                  host hands over (win, step) accordingly.  SAME: out = ceil(H / step), pad_before = pad_total / 2,
//                  padded cells never win.  Gradient: to the FIRST maximum of each window in row-major order (TensorFlow's
//                  CPU MaxPoolGrad); overlapping windows add up.
//   GlobalMaxPool  tf.reduce_max over the spatial dims; gradient indicator(x == max) / count (ties share, as
//                  TensorFlow's _MinOrMaxGrad does): `cnt` [n][C] carries the count from the forward pass.
//
// Max-pooling commutes with ReLU, so the engine pools the stored PRE-activation map and keeps applying ReLU on load:
// where the pooled value is positive the arg-max is the same element, where it is not the gradient is masked anyway.
// HBM-bound elementwise kernels, one thread per output (forward) / input (backward) element; no atomics: the
// backward GATHERS from the windows that cover an input element, so the result does not depend on scheduling.
#include "common.h"

struct PoolP {
    const float *x;  float *y;  float *cnt;  const float *dy;  float *dx;
    int n, H, W, C, Ho, Wo, win, step, py, px, global;
};

__global__ __launch_bounds__(256) void pool_fwd_k(const PoolP p) {
    const long total = (long)p.n * p.Ho * p.Wo * p.C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % p.C);
        long r = e / p.C;
        const int ox = (int)(r % p.Wo);  r /= p.Wo;
        const int oy = (int)(r % p.Ho);
        const int n = (int)(r / p.Ho);
        const int y0 = p.global ? 0 : oy * p.step - p.py, x0 = p.global ? 0 : ox * p.step - p.px;
        const int wy = p.global ? p.H : p.win, wx = p.global ? p.W : p.win;
        float m = -INFINITY;
        int k = 0;
        for (int dy = 0; dy < wy; ++dy)
            for (int dx = 0; dx < wx; ++dx) {
                const int iy = y0 + dy, ix = x0 + dx;
                if ((unsigned)iy >= (unsigned)p.H || (unsigned)ix >= (unsigned)p.W) continue;
                const float v = p.x[(((size_t)n * p.H + iy) * p.W + ix) * p.C + c];
                if (v > m) { m = v; k = 1; } else if (v == m) ++k;
            }
        p.y[e] = m;
        if (p.global && p.cnt) p.cnt[e] = (float)k;
    }
}

__global__ __launch_bounds__(256) void pool_bwd_k(const PoolP p) {
    const long total = (long)p.n * p.H * p.W * p.C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % p.C);
        long r = e / p.C;
        const int ix = (int)(r % p.W);  r /= p.W;
        const int iy = (int)(r % p.H);
        const int n = (int)(r / p.H);
        const float v = p.x[e];
        float g = 0.f;
        if (p.global) {
            const size_t o = (size_t)n * p.C + c;
            if (v == p.y[o]) g = p.dy[o] / p.cnt[o];
        } else {
            // windows (oy, ox) that contain (iy, ix):  oy * step - py <= iy < oy * step - py + win
            const int oy_hi = (iy + p.py) / p.step, ox_hi = (ix + p.px) / p.step;
            for (int oy = oy_hi; oy >= 0 && oy * p.step - p.py + p.win > iy; --oy) {
                if (oy >= p.Ho) continue;
                for (int ox = ox_hi; ox >= 0 && ox * p.step - p.px + p.win > ix; --ox) {
                    if (ox >= p.Wo) continue;
                    const size_t o = (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.C + c;
                    if (v != p.y[o]) continue;
                    // the FIRST maximum of the window in row-major order gets the gradient
                    const int y0 = oy * p.step - p.py, x0 = ox * p.step - p.px;
                    bool first = true;
                    for (int dy = 0; dy < p.win && first; ++dy)
                        for (int dx = 0; dx < p.win; ++dx) {
                            const int qy = y0 + dy, qx = x0 + dx;
                            if (qy == iy && qx == ix) { dy = p.win; break; }              // reached this element: nothing earlier ties
                            if ((unsigned)qy >= (unsigned)p.H || (unsigned)qx >= (unsigned)p.W) continue;
                            if (p.x[(((size_t)n * p.H + qy) * p.W + qx) * p.C + c] == v) { first = false; break; }
                        }
                    if (first) g += p.dy[o];
                }
            }
        }
        p.dx[e] = g;
    }
}

static int pool_fill(PoolP &p, int n, int H, int W, int C, int win, int step, int global) {
    if (n <= 0 || H <= 0 || W <= 0 || C <= 0) return MPNN_E_ARG;
    if (!global && (win < 1 || step < 1)) return MPNN_E_ARG;
    p.n = n; p.H = H; p.W = W; p.C = C; p.win = win; p.step = step; p.global = global ? 1 : 0;
    if (global) { p.Ho = p.Wo = 1; p.py = p.px = 0; return 0; }
    p.Ho = (H + step - 1) / step;  p.Wo = (W + step - 1) / step;
    const int ty = (p.Ho - 1) * step + win - H, tx = (p.Wo - 1) * step + win - W;
    p.py = ty > 0 ? ty / 2 : 0;  p.px = tx > 0 ? tx / 2 : 0;
    return 0;
}

static unsigned pool_grid(long total) {
    long b = (total + 1023) / 1024;
    return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

extern "C" int mpnn_maxpool_fwd(const float *x, float *y, float *cnt, int n, int H, int W, int C, int win, int step,
                                int global, void *stream) {
    if (n <= 0) return 0;
    if (!x || !y || (global && !cnt)) return MPNN_E_ARG;
    PoolP p = {};
    const int rc = pool_fill(p, n, H, W, C, win, step, global);
    if (rc) return rc;
    p.x = x; p.y = y; p.cnt = cnt;
    hipLaunchKernelGGL(pool_fwd_k, dim3(pool_grid((long)n * p.Ho * p.Wo * C)), dim3(256), 0, (hipStream_t)stream, p);
    MPNN_LAUNCH_CHECK();
    return 0;
}

extern "C" int mpnn_maxpool_bwd(const float *x, const float *y, const float *cnt, const float *dy, float *dx, int n, int H,
                                int W, int C, int win, int step, int global, void *stream) {
    if (n <= 0) return 0;
    if (!x || !y || !dy || !dx || (global && !cnt)) return MPNN_E_ARG;
    PoolP p = {};
    const int rc = pool_fill(p, n, H, W, C, win, step, global);
    if (rc) return rc;
    p.x = x; p.y = const_cast<float *>(y); p.cnt = const_cast<float *>(cnt); p.dy = dy; p.dx = dx;
    hipLaunchKernelGGL(pool_bwd_k, dim3(pool_grid((long)n * H * W * C)), dim3(256), 0, (hipStream_t)stream, p);
    MPNN_LAUNCH_CHECK();
    return 0;
}
