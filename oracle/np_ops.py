"""Float64 NumPy restatement of every array op on the multipath-nn hot path.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  PARITY UNPINNED: TensorFlow
cannot run here; each function states the TensorFlow semantics it assumes and
cites the reference line that calls it (paths relative to /root/reference).

Layout conventions follow the reference: activations NHWC, filters HWIO,
row-major flattening.  Every forward has a hand-derived backward so the HIP
kernels can be checked one at a time; the backward formulas themselves are
cross-checked against torch autograd in ``tests/test_oracle_vs_torch.py``.
"""
import numpy as np

F = np.float64


# ----------------------------------------------------------------------------
# ToPyramid -- scripts/lib/layer_types.py:118-125
# tf.image.resize_images (legacy bilinear, align_corners=False) at an integer
# ratio r maps src = dst * r exactly, i.e. a strided pick.
# ----------------------------------------------------------------------------
def pyramid(x, n_scales):
    return [np.ascontiguousarray(x[:, ::2 ** i, ::2 ** i, :]) for i in range(n_scales)]


# ----------------------------------------------------------------------------
# conv / pool helpers -- scripts/lib/layer_types.py:106-110
# tf.nn.conv2d stride 1 'SAME' = cross-correlation, pad (k-1)//2 before and
# k-1-(k-1)//2 after.  tf.nn.max_pool 2x2/2 'SAME' on even maps = no padding.
# ----------------------------------------------------------------------------
def _pad_same(x, kh, kw):
    pt, pl = (kh - 1) // 2, (kw - 1) // 2
    return np.pad(x, ((0, 0), (pt, kh - 1 - pt), (pl, kw - 1 - pl), (0, 0)))


def conv_same(x, w):
    x = np.asarray(x, F); w = np.asarray(w, F)
    n, h, wd, _ = x.shape
    kh, kw, _, co = w.shape
    xp = _pad_same(x, kh, kw)
    out = np.zeros((n, h, wd, co), F)
    for i in range(kh):
        for j in range(kw):
            out += xp[:, i:i + h, j:j + wd, :] @ w[i, j]
    return out


def conv_same_bwd(x, w, gy):
    """Returns (dL/dx, dL/dw) for y = conv_same(x, w)."""
    x = np.asarray(x, F); w = np.asarray(w, F); gy = np.asarray(gy, F)
    n, h, wd, ci = x.shape
    kh, kw, _, co = w.shape
    pt, pl = (kh - 1) // 2, (kw - 1) // 2
    xp = _pad_same(x, kh, kw)
    gxp = np.zeros_like(xp)
    gw = np.zeros_like(w)
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, i:i + h, j:j + wd, :]
            gw[i, j] = np.tensordot(patch, gy, axes=([0, 1, 2], [0, 1, 2]))
            gxp[:, i:i + h, j:j + wd, :] += gy @ w[i, j].T
    return gxp[:, pt:pt + h, pl:pl + wd, :], gw


def pool2(x):
    x = np.asarray(x, F)
    n, h, w, c = x.shape
    return x.reshape(n, h // 2, 2, w // 2, 2, c).max(axis=(2, 4))


def pool2_argfirst(x):
    """Index 0..3 (row-major in the 2x2 window) of the FIRST maximum.
    Assumed TF CPU MaxPoolGrad tie-break (assumption 3)."""
    x = np.asarray(x, F)
    n, h, w, c = x.shape
    win = x.reshape(n, h // 2, 2, w // 2, 2, c).transpose(0, 1, 3, 2, 4, 5)
    win = win.reshape(n, h // 2, w // 2, 4, c)
    return np.argmax(win, axis=3)          # np.argmax = first occurrence


def pool2_bwd(x, gy):
    n, h, w, c = x.shape
    arg = pool2_argfirst(x)
    gwin = np.zeros((n, h // 2, w // 2, 4, c), F)
    np.put_along_axis(gwin, arg[:, :, :, None, :], np.asarray(gy, F)[:, :, :, None, :], axis=3)
    gwin = gwin.reshape(n, h // 2, w // 2, 2, 2, c).transpose(0, 1, 3, 2, 4, 5)
    return gwin.reshape(n, h, w, c)


# ----------------------------------------------------------------------------
# MultiscaleConvMax -- scripts/lib/layer_types.py:149-194
# ----------------------------------------------------------------------------
def msconv_fwd(xs, w_horz, w_vert, b):
    """xs: the LAST len(b) scales of the input pyramid (negative indexing,
    layer_types.py:163,181-185).  Returns the list of pre-BN sums."""
    L = len(b)
    xs = xs[-L:]
    out = [b[0] + conv_same(xs[0], w_horz[0])]
    for i in range(1, L):
        out.append(b[i] + conv_same(xs[i], w_horz[i])
                   + conv_same(pool2(out[i - 1]), w_vert[i - 1]))
    return out


def msconv_bwd(xs, w_horz, w_vert, b, out, g_out):
    """g_out[i] = dL/d out[i] from BatchNorm only; returns
    (dxs, dw_horz, dw_vert, db, g_total) where g_total includes the vert path."""
    L = len(b)
    xs = xs[-L:]
    g = [np.array(gi, F) for gi in g_out]
    dxs, dwh, dwv, db = [None] * L, [None] * L, [None] * (L - 1), [None] * L
    for i in range(L - 1, -1, -1):
        db[i] = g[i].sum(axis=(0, 1, 2))
        dxs[i], dwh[i] = conv_same_bwd(xs[i], w_horz[i], g[i])
        if i > 0:
            pin = pool2(out[i - 1])
            dp, dwv[i - 1] = conv_same_bwd(pin, w_vert[i - 1], g[i])
            g[i - 1] = g[i - 1] + pool2_bwd(out[i - 1], dp)
    return dxs, dwh, dwv, db, g


def msconv_n_ops(out_hw, w_horz_shapes, w_vert_shapes):
    """layer_types.py:189-194."""
    tot = 0
    for i, (h, w) in enumerate(out_hw):
        tot += h * w * (int(np.prod(w_horz_shapes[i]))
                        + (int(np.prod(w_vert_shapes[i - 1])) if i > 0 else 0))
    return tot


# ----------------------------------------------------------------------------
# BatchNorm -- scripts/lib/layer_types.py:219-239 (d=0.9, eps=1e-6)
# tf.nn.moments = batch mean and BIASED variance over all leading dims.
# ----------------------------------------------------------------------------
def bn_train(x, gamma, beta, eps=1e-6):
    x = np.asarray(x, F)
    ax = tuple(range(x.ndim - 1))
    m = x.mean(axis=ax)
    v = ((x - m) ** 2).mean(axis=ax)
    y = gamma * (x - m) / np.sqrt(v + eps) + beta
    return y, m, v


def bn_moving(m_avg, v_avg, m, v, d=0.9):
    return d * m_avg + (1 - d) * m, d * v_avg + (1 - d) * v


def bn_eval(x, gamma, beta, m_avg, v_avg, eps=1e-6):
    return gamma * (np.asarray(x, F) - m_avg) / np.sqrt(v_avg + eps) + beta


def bn_train_bwd(x, gamma, m, v, gy, eps=1e-6):
    """Gradient THROUGH the batch statistics.  Returns (gx, dgamma, dbeta)."""
    x = np.asarray(x, F); gy = np.asarray(gy, F)
    ax = tuple(range(x.ndim - 1))
    cnt = x.size // x.shape[-1]
    rstd = 1.0 / np.sqrt(v + eps)
    xh = (x - m) * rstd
    dbeta = gy.sum(axis=ax)
    dgamma = (gy * xh).sum(axis=ax)
    gx = gamma * rstd * (gy - dbeta / cnt - xh * dgamma / cnt)
    return gx, dgamma, dbeta


def relu(x):
    return np.maximum(np.asarray(x, F), 0.0)


def relu_bwd(y, gy):
    return np.where(np.asarray(y) > 0, gy, 0.0)


# ----------------------------------------------------------------------------
# LinTrans / Softmax / CrossEntropyError
# scripts/lib/layer_types.py:39-53, 81-84, 262-272
# ----------------------------------------------------------------------------
def lintrans(x, w, b):
    x = np.asarray(x, F)
    return x.reshape(x.shape[0], -1) @ w + b


def lintrans_bwd(x, w, gy):
    x2 = np.asarray(x, F).reshape(x.shape[0], -1)
    return (gy @ w.T).reshape(x.shape), x2.T @ gy, gy.sum(axis=0)


def softmax(z):
    z = np.asarray(z, F)
    e = np.exp(z - z.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


def softmax_bwd(p, gp):
    return p * (gp - (gp * p).sum(axis=1, keepdims=True))


def xent_eps(p, y, eps=1e-6):
    """c_err, delta_cor of CrossEntropyError; argmax = first index on ties."""
    n_cls = y.shape[1]
    q = eps / n_cls + (1 - eps) * p
    c_err = -(y * np.log(q)).sum(axis=1)
    d_cor = (np.argmax(p, 1) == np.argmax(y, 1)).astype(F)
    return c_err, d_cor


def xent_eps_bwd(p, y, g_cerr, eps=1e-6):
    n_cls = y.shape[1]
    q = eps / n_cls + (1 - eps) * p
    return g_cerr[:, None] * (-(y / q) * (1 - eps))


# ----------------------------------------------------------------------------
# Momentum + TALR -- scripts/lib/net_types.py:24-37 and tf.train.MomentumOptimizer
# (non-Nesterov: accum = mu*accum + g ; var -= lr*accum)
# ----------------------------------------------------------------------------
def momentum_step(theta, accum, grad, lr, mu, scale=1.0):
    accum = mu * accum + scale * grad
    return theta - lr * accum, accum


# ----------------------------------------------------------------------------
# data.py augmentation restated (scripts/lib/data.py:10-22) -- pinned by KA7 and
# by fixtures generated from the reference module itself.
# ----------------------------------------------------------------------------
def shift_fill_mean(a, du, dv):
    """b[u, v] = a[u+du, v+dv]; out-of-range filled with the per-channel mean."""
    h, w = a.shape[:2]
    b = np.empty_like(a)
    b[:] = a.mean(axis=(0, 1))
    ua = slice(max(du, 0), min(h + du, h)); va = slice(max(dv, 0), min(w + dv, w))
    ub = slice(max(-du, 0), min(h - du, h)); vb = slice(max(-dv, 0), min(w - dv, w))
    b[ub, vb] = a[ua, va]
    return b


def max_pool_same(x, win, step):
    """tf.nn.max_pool(x, ksize=win, strides=step, 'SAME') on NHWC: (y, arg) with arg = flat index (iy * W + ix) of
    the FIRST maximum of every window in row-major order (TensorFlow's CPU MaxPoolGrad routes the gradient there)."""
    n, H, W, C = x.shape
    ho, wo = -(-H // step), -(-W // step)
    ty, tx = max((ho - 1) * step + win - H, 0), max((wo - 1) * step + win - W, 0)
    py, px = ty // 2, tx // 2
    y = np.full((n, ho, wo, C), -np.inf, x.dtype)
    arg = np.zeros((n, ho, wo, C), np.int64)
    for oy in range(ho):
        for ox in range(wo):
            for dy in range(win):
                for dx in range(win):
                    iy, ix = oy * step - py + dy, ox * step - px + dx
                    if 0 <= iy < H and 0 <= ix < W:
                        v = x[:, iy, ix, :]
                        better = v > y[:, oy, ox, :]
                        y[:, oy, ox, :] = np.where(better, v, y[:, oy, ox, :])
                        arg[:, oy, ox, :] = np.where(better, iy * W + ix, arg[:, oy, ox, :])
    return y, arg


def max_pool_same_bwd(dy, arg, shape):
    n, H, W, C = shape
    dx = np.zeros((n, H * W, C), dy.dtype)
    ni, ci = np.meshgrid(np.arange(n), np.arange(C), indexing='ij')
    for oy in range(dy.shape[1]):
        for ox in range(dy.shape[2]):
            np.add.at(dx, (ni, arg[:, oy, ox, :], ci), dy[:, oy, ox, :])
    return dx.reshape(shape)


def global_max_pool(x):
    """tf.reduce_max over H, W: (y, cnt) with cnt = number of maxima (the gradient is shared among them)."""
    y = x.max(axis=(1, 2))
    return y, (x == y[:, None, None, :]).sum(axis=(1, 2)).astype(x.dtype)
