"""The float64 NumPy oracle vs. an independent implementation (torch-CPU operators and
autograd).  This is what pins the hand-derived forward/backward formulas, since the
reference's TensorFlow cannot run here (oracle/__init__.py: parity unpinned)."""
import numpy as np
import torch
import torch.nn.functional as TF

from oracle import np_ops as O

T = lambda a: torch.tensor(np.asarray(a, np.float64), requires_grad=True)
nhwc = lambda t: t.permute(0, 2, 3, 1)
nchw = lambda t: t.permute(0, 3, 1, 2)


def t_conv(x, w):          # x NHWC, w HWIO
    return nhwc(TF.conv2d(nchw(x), w.permute(3, 2, 0, 1), padding=1))


def t_pool(x):
    return nhwc(TF.max_pool2d(nchw(x), 2))


def close(a, b, tol=1e-10):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    assert np.abs(a - b).max() <= tol * (1 + np.abs(b).max())


def test_conv_fwd_bwd():
    rng = np.random.default_rng(0)
    x, w, gy = rng.standard_normal((3, 8, 8, 5)), rng.standard_normal((3, 3, 5, 7)), rng.standard_normal((3, 8, 8, 7))
    xt, wt = T(x), T(w)
    y = t_conv(xt, wt)
    close(O.conv_same(x, w), y)
    y.backward(torch.tensor(gy))
    gx, gw = O.conv_same_bwd(x, w, gy)
    close(gx, xt.grad); close(gw, wt.grad)


def test_pool_fwd_bwd_and_first_max_ties():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 8, 8, 3)); gy = rng.standard_normal((2, 4, 4, 3))
    xt = T(x); y = t_pool(xt)
    close(O.pool2(x), y)
    y.backward(torch.tensor(gy)); close(O.pool2_bwd(x, gy), xt.grad)
    # ties: the gradient goes to the FIRST maximum in row-major window order
    xt = np.zeros((1, 2, 2, 1)); g = O.pool2_bwd(xt, np.ones((1, 1, 1, 1)))
    assert g.reshape(-1).tolist() == [1.0, 0.0, 0.0, 0.0]


def test_bn_train_fwd_bwd():
    rng = np.random.default_rng(2)
    x, gamma, beta, gy = rng.standard_normal((4, 6, 6, 5)) * 2 + 1, rng.uniform(0.5, 2, 5), rng.standard_normal(5), rng.standard_normal((4, 6, 6, 5))
    xt, gt, bt = T(x), T(gamma), T(beta)
    m = xt.mean((0, 1, 2)); v = ((xt - m) ** 2).mean((0, 1, 2))
    y = gt * (xt - m) / torch.sqrt(v + 1e-6) + bt
    yo, mo, vo = O.bn_train(x, gamma, beta)
    close(yo, y); close(mo, m); close(vo, v)
    y.backward(torch.tensor(gy))
    gx, dg, db = O.bn_train_bwd(x, gamma, mo, vo, gy)
    close(gx, xt.grad, 1e-9); close(dg, gt.grad, 1e-9); close(db, bt.grad, 1e-9)
    # torch's own batch_norm (biased variance in training mode) agrees
    ref = nhwc(TF.batch_norm(nchw(torch.tensor(x)), None, None, torch.tensor(gamma), torch.tensor(beta), True, 0.1, 1e-6))
    close(yo, ref, 1e-9)


def test_msconv_block_fwd_bwd():
    rng = np.random.default_rng(3)
    n, chans = 2, [4, 6, 8]
    xs = [rng.standard_normal((n, 16 >> i, 16 >> i, 3)) for i in range(4)]      # 4 scales in, 3 used
    wh = [rng.standard_normal((3, 3, 3, c)) * 0.3 for c in chans]
    wv = [rng.standard_normal((3, 3, chans[i], chans[i + 1])) * 0.3 for i in range(2)]
    b = [rng.standard_normal(c) for c in chans]
    g_out = [rng.standard_normal((n, 8 >> i, 8 >> i, c)) for i, c in enumerate(chans)]
    xt, wht, wvt, bt = [T(a) for a in xs], [T(a) for a in wh], [T(a) for a in wv], [T(a) for a in b]
    xl = xt[-3:]
    out = [bt[0] + t_conv(xl[0], wht[0])]
    for i in range(1, 3):
        out.append(bt[i] + t_conv(xl[i], wht[i]) + t_conv(t_pool(out[i - 1]), wvt[i - 1]))
    o_np = O.msconv_fwd(xs, wh, wv, b)
    for a, c in zip(o_np, out):
        close(a, c)
    sum((o * torch.tensor(g)).sum() for o, g in zip(out, g_out)).backward()
    dxs, dwh, dwv, db, _ = O.msconv_bwd(xs, wh, wv, b, o_np, g_out)
    for i in range(3):
        close(dxs[i], xl[i].grad, 1e-9); close(dwh[i], wht[i].grad, 1e-9); close(db[i], bt[i].grad, 1e-9)
    for i in range(2):
        close(dwv[i], wvt[i].grad, 1e-9)


def test_lintrans_softmax_xent():
    rng = np.random.default_rng(4)
    x, w, b = rng.standard_normal((6, 2, 2, 5)), rng.standard_normal((20, 4)), rng.standard_normal(4)
    y = np.eye(4)[rng.integers(0, 4, 6)]; gc = rng.standard_normal(6)
    xt, wt, bt = T(x), T(w), T(b)
    z = xt.reshape(6, -1) @ wt + bt
    p = torch.softmax(z, 1)
    ce = -(torch.tensor(y) * torch.log(1e-6 / 4 + (1 - 1e-6) * p)).sum(1)
    zo = O.lintrans(x, w, b); po = O.softmax(zo); co, dcor = O.xent_eps(po, y)
    close(zo, z); close(po, p); close(co, ce)
    (ce * torch.tensor(gc)).sum().backward()
    gz = O.softmax_bwd(po, O.xent_eps_bwd(po, y, gc))
    gx, gw, gb = O.lintrans_bwd(x, w, gz)
    close(gx, xt.grad, 1e-9); close(gw, wt.grad, 1e-9); close(gb, bt.grad, 1e-9)


def test_pyramid_is_strided_pick():
    x = np.arange(2 * 8 * 8 * 3, dtype=np.float64).reshape(2, 8, 8, 3)
    p = O.pyramid(x, 4)
    assert [a.shape for a in p] == [(2, 8, 8, 3), (2, 4, 4, 3), (2, 2, 2, 3), (2, 1, 1, 3)]
    assert np.array_equal(p[2], x[:, ::4, ::4, :])
