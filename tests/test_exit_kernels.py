"""Kernel-level parity of the exit path and the optimizer, through the C ABI, each entry point alone
against a float64 oracle (numpy / torch-CPU autograd):

  mpnn_lin_fwd / mpnn_lin_bwd          LinTrans of head + router over relu(bn(x))   layer_types.py:39-53
  mpnn_exit_tail_fwd / _bwd            Softmax + CrossEntropyError, router tail      :81-84,262-272; arch_and_hypers.py:47-49
  mpnn_exit_ev                         the same in evaluation mode + routing lists   net_types.py:127-129
  mpnn_route                           actor / critic / SR on 2-, 3- and 4-way trees net_types.py:108-131,165-177,193-243
  mpnn_talr_momentum_step              TALR + L2 + momentum                           net_types.py:24-37
  mpnn_bn_relu_fwd                     the materialised activation

Tolerances: 2e-5 * (1 + max|ref|) for forward values, 1e-4 * max|ref| for gradients (fp32
accumulation over <= 2048 terms against float64); discrete outputs exact.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lib import _hip
from hiputil import DEV, dev, stream, bn_dict, unslot


def close(a, b, tol, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b).max() if a.size else 0.0
    assert err <= tol * (1 + np.abs(b).max()), (what, err, np.abs(b).max())


def gclose(a, b, what, tol=1e-4, floor=1e-6):
    """floor: some gradients are analytically zero (a bias ahead of a BatchNorm) and come out as
    fp32 rounding noise of the terms that cancel."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b).max()
    assert err <= tol * np.abs(b).max() + floor, (what, err, np.abs(b).max())


def bn_relu(x, g, b, eps=1e-6):
    x = np.asarray(x, np.float64)
    c = x.shape[-1]
    f = x.reshape(-1, c)
    m, v = f.mean(0), f.var(0)
    xh = (x - m) / np.sqrt(v + eps)
    return np.maximum(g * xh + b, 0), xh


# ---------------------------------------------------------------------------------- lin
# (200, 257: more rows than the 128 the backward holds at a time -- its outer loop, a ragged last pass)
@pytest.mark.parametrize('n,C_,dyn', [(37, 16, False), (128, 128, False), (16, 64, True), (5, 32, True), (200, 32, True), (257, 16, False)])
def test_lin_fwd_bwd(n, C_, dyn):
    lib = _hip.load()
    rng = np.random.default_rng(n + C_)
    HW, M0, M1 = 16, 10, 16
    K = HW * C_
    x = rng.standard_normal((n, 4, 4, C_)).astype(np.float32)
    g, be = rng.random(C_).astype(np.float32) + 0.5, (rng.standard_normal(C_) * 0.2).astype(np.float32)
    w0 = (rng.standard_normal((K, M0)) / np.sqrt(K)).astype(np.float32)
    w1 = (rng.standard_normal((K + (1 if dyn else 0), M1)) / np.sqrt(K)).astype(np.float32)
    b0, b1 = rng.standard_normal(M0).astype(np.float32), rng.standard_normal(M1).astype(np.float32)
    kc = rng.choice([0.0, 1e-9, 6.4e-8], n).astype(np.float32)
    alpha = 1e7
    bn, cnt = bn_dict(x, g, be)
    xd, w0d, w1d, b0d, b1d, kcd = dev(x), dev(w0), dev(w1), dev(b0), dev(b1), dev(kc)
    y0, y1 = torch.empty((n, M0), device=DEV), torch.empty((n, M1), device=DEV)
    lf = _hip.LinFwdArgs()
    lf.a = _hip.act(xd, C_, _hip.ACT_BN_BATCH, 0, bn, cnt)
    lf.HW, lf.n = HW, n
    lf.w[0], lf.b[0], lf.y[0], lf.M[0] = w0d.data_ptr(), b0d.data_ptr(), y0.data_ptr(), M0
    lf.w[1], lf.b[1], lf.y[1], lf.M[1] = w1d.data_ptr(), b1d.data_ptr(), y1.data_ptr(), M1
    lf.k_cpt, lf.alpha_cpt = kcd.data_ptr(), alpha
    lf.extra_col[1] = 1 if dyn else 0
    tab = _hip.to_device_table([lf], DEV)
    _hip.check(lib.mpnn_lin_fwd(tab.data_ptr(), 1, n, stream()), 'lin_fwd')
    torch.cuda.synchronize()
    a, xh = bn_relu(x, g, be)
    af = a.reshape(n, K)
    r0 = af @ w0.astype(np.float64) + b0
    r1 = af @ w1[:K].astype(np.float64) + b1 + (alpha * kc[:, None].astype(np.float64) * w1[K] if dyn else 0)
    close(y0.cpu().numpy(), r0, 2e-5, 'head logits')
    close(y1.cpu().numpy(), r1, 2e-5, 'router h1')
    # the same map K-SLICED (scratch given): partial tiles met by the last workgroup to arrive.  Three
    # launches in a row: the ticket counters must be left at zero, and the result may not depend on the
    # arrival order (bit-identical every time).
    rg = (n + 15) // 16
    kpart = torch.full((rg * _hip.LIN_KSLICES * 512,), float('nan'), device=DEV)
    kcnt = torch.zeros(rg, dtype=torch.int32, device=DEV)
    lf.kpart, lf.kcnt = kpart.data_ptr(), kcnt.data_ptr()
    tab2 = _hip.to_device_table([lf], DEV)
    outs = []
    for rep in range(3):
        y0.fill_(7.0); y1.fill_(7.0)
        _hip.check(lib.mpnn_lin_fwd_ks(tab2.data_ptr(), 1, n, K, stream()), 'lin_fwd sliced')
        torch.cuda.synchronize()
        assert int(kcnt.abs().sum()) == 0
        outs.append((y0.cpu().numpy().copy(), y1.cpu().numpy().copy()))
    close(outs[0][0], r0, 2e-5, 'head logits (K-sliced)')
    close(outs[0][1], r1, 2e-5, 'router h1 (K-sliced)')
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1])
    lf.kpart, lf.kcnt = None, None

    # backward: dX, dW, db (accumulated into zeroed tensors); fused BatchNorm-backward reductions
    dy0, dy1 = rng.standard_normal((n, M0)).astype(np.float32), rng.standard_normal((n, M1)).astype(np.float32)
    nblk = (K + 1 + 63) // 64
    bpart = torch.full((nblk * _hip.LIN_RSPLIT * _hip.LIN_RS_TILE,), float('nan'), device=DEV)
    bcnt = torch.zeros(nblk, dtype=torch.int32, device=DEV)
    prev = {}
    # (rs: the ROW-SPLIT form, mpnn_lin_bwd_rs -- partial tiles met by the last row group to arrive; run
    # twice: the ticket counters must be left at zero and the result may not depend on the arrival order)
    for fused, rs in ((False, False), (True, False), (False, True), (True, True), (True, True)):
        dy0d, dy1d = dev(dy0), dev(dy1)
        dw0, dw1 = torch.zeros_like(w0d), torch.zeros_like(w1d)
        db0, db1 = torch.zeros(M0, device=DEV), torch.zeros(M1, device=DEV)
        dx = torch.full((n, K), 9.0, device=DEV)
        dz = torch.full((n, K), 9.0, device=DEV)
        red = torch.zeros(_hip.BN_SLOTS * 2 * C_, device=DEV, dtype=torch.float64)
        lb = _hip.LinBwdArgs()
        lb.a, lb.HW, lb.n = lf.a, HW, n
        lb.w[0], lb.dy[0], lb.M[0], lb.dw[0], lb.db[0] = w0d.data_ptr(), dy0d.data_ptr(), M0, dw0.data_ptr(), db0.data_ptr()
        lb.w[1], lb.dy[1], lb.M[1], lb.dw[1], lb.db[1] = w1d.data_ptr(), dy1d.data_ptr(), M1, dw1.data_ptr(), db1.data_ptr()
        lb.k_cpt, lb.alpha_cpt = kcd.data_ptr(), alpha
        lb.extra_col[1] = 1 if dyn else 0
        if fused:
            lb.dx, lb.dz_out, lb.red_out, lb.red_nslot = None, dz.data_ptr(), red.data_ptr(), _hip.BN_SLOTS
        else:
            lb.dx = dx.data_ptr()
        if rs:
            lb.kpart, lb.kcnt = bpart.data_ptr(), bcnt.data_ptr()
        tb = _hip.to_device_table([lb], DEV)
        _hip.check((lib.mpnn_lin_bwd_rs if rs else lib.mpnn_lin_bwd)(tb.data_ptr(), 1, n, K, stream()), 'lin_bwd')
        torch.cuda.synchronize()
        if rs:
            assert int(bcnt.abs().sum()) == 0
            got = [t.cpu().numpy().copy() for t in (dw0, dw1, db0, db1, dz if fused else dx)]
            if fused in prev:
                assert all(np.array_equal(u, v) for u, v in zip(got, prev[fused])), 'row-split result depends on arrival order'
            prev[fused] = got
        dxr = dy0.astype(np.float64) @ w0.T + dy1.astype(np.float64) @ w1[:K].T
        gclose(dw0.cpu().numpy(), af.T @ dy0, 'dW head')
        ext = np.concatenate([af, alpha * kc[:, None].astype(np.float64)], 1) if dyn else af
        gclose(dw1.cpu().numpy(), ext.T @ dy1, 'dW router')
        gclose(db0.cpu().numpy(), dy0.sum(0, dtype=np.float64), 'db head')
        gclose(db1.cpu().numpy(), dy1.sum(0, dtype=np.float64), 'db router')
        if not fused:
            gclose(dx.cpu().numpy(), dxr, 'dX')
        else:
            dzr = dxr * (af > 0)
            gclose(dz.cpu().numpy(), dzr, 'dz (masked dX)')
            dz4 = dzr.reshape(n, 4, 4, C_)
            gclose(unslot(red, 2 * C_), np.concatenate([dz4.sum((0, 1, 2)), (dz4 * xh).sum((0, 1, 2))]), 'BN reductions')


# ---------------------------------------------------------------------------------- exit tail
def tail_ref(z, y, h1, P, eps_ce, bn_eps, w_cerr, dr, S, moving=None):
    T = lambda a, g=False: torch.tensor(np.asarray(a, np.float64), requires_grad=g)
    zt, h1t = T(z, True), T(h1, True)
    p = {k: T(v, True) for k, v in P.items()}
    yt = T(y)
    nc = z.shape[1]
    sm = torch.softmax(zt, 1)
    c_err = -(yt * torch.log(eps_ce / nc + (1 - eps_ce) * sm)).sum(1)
    d_cor = (torch.argmax(sm, 1) == torch.argmax(yt, 1)).double()
    stats = []

    def bn(x, g, b, k):
        if moving is None:
            m, v = x.mean(0), ((x - x.mean(0)) ** 2).mean(0)
        else:
            m, v = T(moving[2 * k]), T(moving[2 * k + 1])
        stats.append((m.detach().numpy(), v.detach().numpy()))
        return g * (x - m) / torch.sqrt(v + bn_eps) + b
    a1 = torch.relu(bn(h1t, p['g1'], p['b1'], 0))
    h2 = a1 @ p['w2'] + p['bias2']
    a2 = torch.relu(bn(h2, p['g2'], p['b2'], 1))
    r = a2 @ p['w3'] + p['bias3']
    out = dict(c_err=c_err.detach().numpy(), d_cor=d_cor.numpy(), h2=h2.detach().numpy(), r=r.detach().numpy(), stats=stats)
    if w_cerr is not None:
        L = (T(w_cerr) * c_err).sum() + (T(dr) * r).sum()
        L.backward()
        out['dz'], out['dh1'] = zt.grad.numpy(), h1t.grad.numpy()
        out.update({'d' + k: v.grad.numpy() for k, v in p.items()})
    return out


def tail_params(rng, R, S):
    f = np.float32
    return dict(g1=(rng.random(R) + 0.5).astype(f), b1=(rng.standard_normal(R) * 0.3).astype(f),
                w2=(rng.standard_normal((R, R)) / 4).astype(f), bias2=(rng.standard_normal(R) * 0.1).astype(f),
                g2=(rng.random(R) + 0.5).astype(f), b2=(rng.standard_normal(R) * 0.3).astype(f),
                w3=(rng.standard_normal((R, S)) / 4).astype(f), bias3=(rng.standard_normal(S) * 0.1).astype(f))


@pytest.mark.parametrize('n,S,R', [(128, 2, 16), (37, 3, 16), (16, 4, 16), (64, 2, 8), (129, 2, 16), (300, 3, 16), (1000, 2, 8)])
def test_exit_tail_fwd_bwd(n, S, R):
    lib = _hip.load()
    rng = np.random.default_rng(10 * n + S)
    nc = 10
    z = rng.standard_normal((n, nc)).astype(np.float32) * 2
    y = np.eye(nc, dtype=np.float32)[rng.integers(0, nc, n)]
    h1 = rng.standard_normal((n, R)).astype(np.float32)
    P = tail_params(rng, R, S)
    eps_ce, bn_eps, decay = 1e-6, 1e-6, 0.9
    d = {k: dev(v) for k, v in P.items()}
    zd, yd, h1d = dev(z), dev(y), dev(h1)
    m1, v1, m2, v2 = (dev(rng.standard_normal(R) * 0.1), dev(rng.random(R) + 0.5), dev(rng.standard_normal(R) * 0.1), dev(rng.random(R) + 0.5))
    mv0 = [t.cpu().numpy().copy() for t in (m1, v1, m2, v2)]
    MS = 4
    c_err, d_cor = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    h2, r, save = torch.empty((n, R), device=DEV), torch.zeros((n, MS), device=DEV), torch.empty(4 * R, device=DEV)
    tf = _hip.ExitTailArgs()
    tf.z, tf.y, tf.n_cls, tf.eps_ce, tf.c_err, tf.d_cor = zd.data_ptr(), yd.data_ptr(), nc, eps_ce, c_err.data_ptr(), d_cor.data_ptr()
    tf.h1, tf.R, tf.n_sinks = h1d.data_ptr(), R, S
    tf.g1, tf.b1, tf.m1, tf.v1 = d['g1'].data_ptr(), d['b1'].data_ptr(), m1.data_ptr(), v1.data_ptr()
    tf.w2, tf.bias2 = d['w2'].data_ptr(), d['bias2'].data_ptr()
    tf.g2, tf.b2, tf.m2, tf.v2 = d['g2'].data_ptr(), d['b2'].data_ptr(), m2.data_ptr(), v2.data_ptr()
    tf.w3, tf.bias3 = d['w3'].data_ptr(), d['bias3'].data_ptr()
    tf.h2, tf.r, tf.r_stride, tf.bn_save = h2.data_ptr(), r.data_ptr(), MS, save.data_ptr()
    tf.bn_eps, tf.bn_decay, tf.mode, tf.n = bn_eps, decay, _hip.ACT_BN_BATCH, n
    tf.bn_eps2, tf.bn_decay2 = bn_eps, decay
    tab = _hip.to_device_table([tf], DEV)
    _hip.check(lib.mpnn_exit_tail_fwd(tab.data_ptr(), 1, n, stream()), 'exit_tail_fwd')
    torch.cuda.synchronize()
    w_cerr = rng.random(n).astype(np.float32) / n
    dr = (rng.standard_normal((n, S)) / n).astype(np.float32)
    ref = tail_ref(z, y, h1, P, eps_ce, bn_eps, w_cerr, dr, S)
    close(c_err.cpu().numpy(), ref['c_err'], 2e-5, 'c_err')
    assert np.array_equal(d_cor.cpu().numpy(), ref['d_cor'])
    close(h2.cpu().numpy(), ref['h2'], 2e-5, 'h2')
    close(r.cpu().numpy()[:, :S], ref['r'], 2e-5, 'r')
    sv = save.cpu().numpy()
    (mm1, vv1), (mm2, vv2) = ref['stats']
    close(sv[:R], mm1, 2e-5, 'bn1 mean'); close(sv[R:2 * R], 1 / np.sqrt(vv1 + bn_eps), 2e-5, 'bn1 rstd')
    close(sv[2 * R:3 * R], mm2, 2e-5, 'bn2 mean'); close(sv[3 * R:], 1 / np.sqrt(vv2 + bn_eps), 2e-5, 'bn2 rstd')
    for t, old, new in ((m1, mv0[0], mm1), (v1, mv0[1], vv1), (m2, mv0[2], mm2), (v2, mv0[3], vv2)):
        close(t.cpu().numpy(), decay * old + (1 - decay) * new, 2e-5, 'moving average')   # layer_types.py:233-234

    drp = np.zeros((n, MS), np.float32); drp[:, :S] = dr
    tb = _hip.ExitTailBwdArgs()
    tb.f = tf
    g = {k: torch.full(v.shape, 9.0, device=DEV) for k, v in P.items()}
    dz, dh1 = torch.empty((n, nc), device=DEV), torch.empty((n, R), device=DEV)
    wcd, drd = dev(w_cerr), dev(drp)
    tb.w_cerr, tb.dr, tb.dz, tb.dh1 = wcd.data_ptr(), drd.data_ptr(), dz.data_ptr(), dh1.data_ptr()
    tb.dg1, tb.db1, tb.dw2, tb.dbias2 = g['g1'].data_ptr(), g['b1'].data_ptr(), g['w2'].data_ptr(), g['bias2'].data_ptr()
    tb.dg2, tb.db2, tb.dw3, tb.dbias3 = g['g2'].data_ptr(), g['b2'].data_ptr(), g['w3'].data_ptr(), g['bias3'].data_ptr()
    tbb = _hip.to_device_table([tb], DEV)
    _hip.check(lib.mpnn_exit_tail_bwd(tbb.data_ptr(), 1, n, stream()), 'exit_tail_bwd')
    torch.cuda.synchronize()
    gclose(dz.cpu().numpy(), ref['dz'], 'dz')
    gclose(dh1.cpu().numpy(), ref['dh1'], 'dh1')
    for k in P:
        gclose(g[k].cpu().numpy(), ref['d' + k], 'd' + k)


# ---------------------------------------------------------------------------------- exit_ev
@pytest.mark.parametrize('n,S,C_,dyn', [(1000, 2, 128, False), (77, 3, 64, True), (16, 4, 16, False)])
def test_exit_ev_against_oracle_with_lists(n, S, C_, dyn):
    lib = _hip.load()
    rng = np.random.default_rng(n + S)
    HW, nc, R, MS = 16, 10, 16, 4
    K = HW * C_
    N = n + 40                                        # images in the buffers; the node's list holds n of them
    x = rng.standard_normal((N, 4, 4, C_)).astype(np.float32)
    g, be = rng.random(C_).astype(np.float32) + 0.5, (rng.standard_normal(C_) * 0.2).astype(np.float32)
    ma, va = (rng.standard_normal(C_) * 0.2).astype(np.float32), (rng.random(C_) + 0.5).astype(np.float32)
    wh = (rng.standard_normal((K, nc)) / np.sqrt(K)).astype(np.float32)
    bh = rng.standard_normal(nc).astype(np.float32)
    w1 = (rng.standard_normal((K + (1 if dyn else 0), R)) / np.sqrt(K)).astype(np.float32)
    b1 = rng.standard_normal(R).astype(np.float32)
    P = tail_params(rng, R, S)
    mov = [(rng.standard_normal(R) * 0.2).astype(np.float32), (rng.random(R) + 0.5).astype(np.float32),
           (rng.standard_normal(R) * 0.2).astype(np.float32), (rng.random(R) + 0.5).astype(np.float32)]
    y = np.eye(nc, dtype=np.float32)[rng.integers(0, nc, N)]
    kc = rng.choice([0.0, 1e-9, 6.4e-8], N).astype(np.float32)
    idx = rng.permutation(N)[:n].astype(np.int32)
    bn = dict(sum=None, gamma=dev(g), beta=dev(be), m_avg=dev(ma), v_avg=dev(va), eps=1e-6)
    d = {k: dev(v) for k, v in P.items()}
    xd, whd, bhd, w1d, b1d, yd, kcd = dev(x), dev(wh), dev(bh), dev(w1), dev(b1), dev(y), dev(kc)
    md = [dev(m) for m in mov]
    idxd, cntd = dev(idx, torch.int32), dev(np.array([n], np.int32), torch.int32)
    c_err, d_cor, r = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV), torch.zeros((N, MS), device=DEV)
    lists = [torch.full((N,), -1, dtype=torch.int32, device=DEV) for _ in range(S)]
    cnts = torch.zeros(4, dtype=torch.int32, device=DEV)
    e = _hip.ExitEvArgs()
    e.a = _hip.act(xd, C_, _hip.ACT_BN_MOVING, 0, bn, 1)
    e.HW, e.n = HW, N
    e.w_head, e.b_head, e.n_cls, e.y, e.eps_ce = whd.data_ptr(), bhd.data_ptr(), nc, yd.data_ptr(), 1e-6
    e.c_err, e.d_cor = c_err.data_ptr(), d_cor.data_ptr()
    e.w1, e.b1, e.R, e.n_sinks = w1d.data_ptr(), b1d.data_ptr(), R, S
    e.extra_col, e.k_cpt, e.alpha_cpt = (1 if dyn else 0), kcd.data_ptr(), 1e7
    e.g1, e.be1, e.m1, e.v1 = d['g1'].data_ptr(), d['b1'].data_ptr(), md[0].data_ptr(), md[1].data_ptr()
    e.w2, e.bias2 = d['w2'].data_ptr(), d['bias2'].data_ptr()
    e.g2, e.be2, e.m2, e.v2 = d['g2'].data_ptr(), d['b2'].data_ptr(), md[2].data_ptr(), md[3].data_ptr()
    e.w3, e.bias3, e.bn_eps, e.bn_eps2 = d['w3'].data_ptr(), d['bias3'].data_ptr(), 1e-6, 1e-6
    e.r, e.r_stride = r.data_ptr(), MS
    e.idx, e.cnt = idxd.data_ptr(), cntd.data_ptr()
    for i in range(1, S):                              # sink 0: the exit's own leaf (no list)
        e.child_idx[i], e.child_cnt[i] = lists[i].data_ptr(), cnts[i:].data_ptr()
    assert lib.mpnn_exit_ev_check(C.byref(e)) == 0
    tab = _hip.to_device_table([e], DEV)
    _hip.check(lib.mpnn_exit_ev(tab.data_ptr(), 1, N, stream()), 'exit_ev')
    torch.cuda.synchronize()
    # oracle on the listed images
    xs = x[idx].astype(np.float64)
    a = np.maximum(g * (xs - ma) / np.sqrt(va.astype(np.float64) + 1e-6) + be, 0).reshape(n, K)
    z = a @ wh + bh
    h1 = a @ w1[:K] + b1 + (1e7 * kc[idx][:, None].astype(np.float64) * w1[K] if dyn else 0)
    ref = tail_ref(z, y[idx], h1, P, 1e-6, 1e-6, None, None, S, moving=mov)
    close(c_err.cpu().numpy()[idx], ref['c_err'], 2e-5, 'c_err')
    assert np.array_equal(d_cor.cpu().numpy()[idx], ref['d_cor'])
    close(r.cpu().numpy()[idx][:, :S], ref['r'], 2e-5, 'r')
    rest = np.setdiff1d(np.arange(N), idx)
    assert not c_err.cpu().numpy()[rest].any() and not r.cpu().numpy()[rest].any()       # untouched
    # lists: the images whose arg-max (first index on ties) is sink i, in any order
    arg = np.argmax(r.cpu().numpy()[idx][:, :S], 1)
    cn = cnts.cpu().numpy()
    for i in range(1, S):
        want = np.sort(idx[arg == i])
        assert cn[i] == len(want)
        assert np.array_equal(np.sort(lists[i].cpu().numpy()[:cn[i]]), want)
    # limits are refused, not truncated
    e.n_cls = 17
    assert lib.mpnn_exit_ev_check(C.byref(e)) == _hip.E_SHAPE


# ---------------------------------------------------------------------------------- route
def chain_tree(depth):
    from oracle.route_ref import Tree
    nodes = [dict(sinks=[1])]                          # pyramid -> block 0
    k = 1
    for dd in range(depth):
        last = dd == depth - 1
        nodes.append(dict(sinks=[k + 1] if last else [k + 1, k + 2]))
        nodes.append(dict(sinks=[]))
        k += 2
    return Tree(nodes)


def mixed_tree():
    """static root -> 3-way switch A {leaf, B, C}; B: 2-way {leaf, static D -> leaf}; C: 4-way {leaf, leaf, E, leaf},
    E static -> 2-way F {leaf, leaf}.  DFS preorder."""
    from oracle.route_ref import Tree
    return Tree([dict(sinks=[1]),                      # 0 root (static)
                 dict(sinks=[2, 3, 7]),                # 1 A
                 dict(),                               # 2 leaf
                 dict(sinks=[4, 5]),                   # 3 B
                 dict(),                               # 4 leaf
                 dict(sinks=[6]),                      # 5 D (static)
                 dict(),                               # 6 leaf
                 dict(sinks=[8, 9, 10, 14]),           # 7 C
                 dict(), dict(),                       # 8, 9 leaves
                 dict(sinks=[11]),                     # 10 E (static)
                 dict(sinks=[12, 13]),                 # 11 F
                 dict(), dict(),                       # 12, 13 leaves
                 dict()])                              # 14 leaf


@pytest.mark.parametrize('kind', ['actor', 'critic', 'critic-opt-cls', 'sr'])
@pytest.mark.parametrize('shape', ['chain8', 'mixed'])
@pytest.mark.parametrize('n', [128, 37])
def test_route_against_oracle(kind, shape, n):
    from oracle.route_ref import route
    lib = _hip.load()
    tree = chain_tree(8) if shape == 'chain8' else mixed_tree()
    rng = np.random.default_rng(len(kind) * 100 + n)
    NN, nl, ns = len(tree.nodes), len(tree.leaves), len(tree.switches)
    MS = max(len(tree.nodes[i]['sinks']) for i in tree.switches)
    rs = [rng.standard_normal((n, len(tree.nodes[i]['sinks']))) * 0.7 for i in tree.switches]
    c_err = rng.random((nl, n)) * 2.5
    d_cor = (rng.random((nl, n)) < 0.5).astype(np.float64)
    ops = rng.integers(1000, 4_000_000, NN).astype(np.float64)
    dyn = n == 37
    k_cpt = rng.choice([0.0, 1e-9, 6.4e-8], n) if dyn else 8e-9
    net_type = {'actor': _hip.NET_ACTOR, 'sr': _hip.NET_SR}.get(kind, _hip.NET_CRITIC)
    okind = 'critic' if kind.startswith('critic') else kind
    opt = kind == 'critic-opt-cls'
    τ, ϵ, k_dec, k_cre = (0.05 if okind == 'critic' else 0.7), 1e-6, 0.01, 1e-3
    ref = route(okind, tree, rs, c_err, d_cor, ops, τ=τ, ϵ=ϵ, k_cpt=k_cpt, k_dec=k_dec, k_cre=k_cre,
                optimistic=opt, use_cls_err=opt)
    tab, kids = tree.tables(MS)
    rbuf = np.zeros((ns, n, MS), np.float32)
    for s, x in enumerate(rs):
        rbuf[s, :, :x.shape[1]] = x
    hyp = np.zeros(_hip.HYP_N, np.float32)
    hyp[_hip.HYP_TAU], hyp[_hip.HYP_EPS], hyp[_hip.HYP_KCPT] = τ, ϵ, (0.0 if dyn else k_cpt)
    hyp[_hip.HYP_KDEC], hyp[_hip.HYP_KCRE] = k_dec, k_cre
    t = dict(tab=dev(tab, torch.int32), kids=dev(kids, torch.int32), ops=dev(ops), hyp=dev(hyp), r=dev(rbuf),
             ce=dev(c_err), dc=dev(d_cor), kv=dev(np.asarray(k_cpt, np.float32) * np.ones(n, np.float32)))
    p_tr, p_ev = torch.empty((NN, n), device=DEV), torch.empty((NN, n), device=DEV)
    w_cerr, drd = torch.zeros((nl, n), device=DEV), torch.zeros((ns, n, MS), device=DEV)
    stat, loss = torch.zeros((NN, 2), device=DEV), torch.zeros(4, device=DEV, dtype=torch.float64)
    ra = _hip.RouteArgs()
    ra.net_type, ra.n_nodes, ra.n_leaves, ra.n_switches, ra.max_sinks = net_type, NN, nl, ns, MS
    ra.optimistic, ra.use_cls_err, ra.want_grad = int(opt), int(opt), 1
    ra.nodes, ra.sw_children, ra.node_ops, ra.hyp = t['tab'].data_ptr(), t['kids'].data_ptr(), t['ops'].data_ptr(), t['hyp'].data_ptr()
    ra.k_cpt_vec = t['kv'].data_ptr() if dyn else None
    ra.r, ra.c_err, ra.d_cor = t['r'].data_ptr(), t['ce'].data_ptr(), t['dc'].data_ptr()
    ra.p_tr, ra.p_ev, ra.w_cerr, ra.dr = p_tr.data_ptr(), p_ev.data_ptr(), w_cerr.data_ptr(), drd.data_ptr()
    ra.node_stat, ra.loss, ra.n, ra.n_total = stat.data_ptr(), loss.data_ptr(), n, n
    _hip.check(lib.mpnn_route(C.byref(ra), stream()), 'route')
    torch.cuda.synchronize()
    close(p_tr.cpu().numpy(), ref['p_tr'], 1e-5, 'p_tr')
    assert np.array_equal(p_ev.cpu().numpy(), ref['p_ev'])
    gclose(w_cerr.cpu().numpy(), ref['w_cerr'], 'dL/dc_err', 2e-5)
    for s, x in enumerate(ref['dr']):
        if np.abs(x).max() > 0:
            gclose(drd.cpu().numpy()[s, :, :x.shape[1]], x, 'dL/dr switch %d' % s, 1e-4)
    lo = loss.cpu().numpy()
    assert abs(lo[3] - n) == 0
    for k in range(3):
        assert abs(lo[k] - ref['loss'][k]) <= 1e-5 * (1e-9 + abs(ref['loss'][k])) + 1e-9, (k, lo[k], ref['loss'][k])
    close(stat.cpu().numpy(), ref['node_stat'], 1e-5, 'TALR node statistics')


# ---------------------------------------------------------------------------------- optimizer
@pytest.mark.parametrize('talr', [0, 1])
def test_talr_momentum_step(talr):
    lib = _hip.load()
    rng = np.random.default_rng(3 + talr)
    sizes = [(5000, 0, 0, 1e-4), (37, 1, 1, 1e-4), (16, 2, 0, 0.0), (2049, 3, 1, 1e-4)]     # (count, node, is_router, l2)
    eq = rng.standard_normal(2049).astype(np.float32)       # the last tensor is a `res` layer: L2 pulls towards w_eq
    eqd = dev(eq)
    n_nodes, n = 4, 128
    total = sum(s[0] for s in sizes)
    P, A, G = (rng.standard_normal(total).astype(np.float32) for _ in range(3))
    p = rng.random((n_nodes, n)) * 0.9 + 0.01
    stat = np.stack([p.sum(1), (p ** 2).sum(1)], 1).astype(np.float32) * 2          # sums over 2 replicas' batches
    lr, mu, artr, world = 0.05, 0.9, 1.7, 2
    hyp = np.zeros(_hip.HYP_N, np.float32)
    hyp[_hip.HYP_LR], hyp[_hip.HYP_MU], hyp[_hip.HYP_ARTR] = lr, mu, artr
    seg, off = [], 0
    want_P, want_A = P.astype(np.float64).copy(), A.astype(np.float64).copy()
    for cnt, node, rt, l2 in sizes:
        has_eq = cnt == 2049
        for s0 in range(0, cnt, 2048):
            seg += [off + s0, min(2048, cnt - s0), node, rt, int(np.float32(l2).view(np.int32)), s0 if has_eq else -1,
                    0, 0, 0, -1, -1, 0]
        pbar = stat[node, 0] / (n * world)
        # net_types.py:25-33: lr_scale = 1/sqrt(mean p_tr^2) with TALR, 1 without; routers get alpha_rtr * lr_scale EITHER WAY
        sc = (1 / np.sqrt(stat[node, 1] / (n * world)) if talr else 1.0) * (artr if rt else 1.0)
        sl = slice(off, off + cnt)
        g = (G[sl].astype(np.float64) / world + 2 * np.float64(np.float32(l2)) * pbar * (P[sl] - (eq if has_eq else 0))) * sc   # net_types.py:24-37 + layer_types.py:52
        want_A[sl] = mu * A[sl] + g
        want_P[sl] = P[sl] - lr * want_A[sl]
        off += cnt
    Pd, Ad, Gd = dev(P), dev(A), dev(G)
    segd, statd, hypd = dev(np.array(seg, np.int32), torch.int32), dev(stat), dev(hyp)
    _hip.check(lib.mpnn_talr_momentum_step(Pd.data_ptr(), Ad.data_ptr(), Gd.data_ptr(), segd.data_ptr(), len(seg) // _hip.SEG_INTS,
                                           statd.data_ptr(), hypd.data_ptr(), talr, 1.0 / (n * world), 1.0 / world,
                                           eqd.data_ptr(), None, stream()), 'talr_momentum_step')
    torch.cuda.synchronize()
    close(Ad.cpu().numpy(), want_A, 1e-6, 'accumulators')
    close(Pd.cpu().numpy(), want_P, 1e-6, 'parameters')


def test_bn_relu_fwd_matches_oracle():
    lib = _hip.load()
    rng = np.random.default_rng(0)
    x = rng.standard_normal((9, 8, 8, 32)).astype(np.float32)
    g, be = rng.random(32).astype(np.float32) + 0.5, rng.standard_normal(32).astype(np.float32) * 0.3
    bn, cnt = bn_dict(x, g, be)
    xd = dev(x)
    y = torch.empty_like(xd)
    a = _hip.act(xd, 32, _hip.ACT_BN_BATCH, 0, bn, cnt)
    _hip.check(lib.mpnn_bn_relu_fwd(C.byref(a), y.data_ptr(), 9 * 64, stream()), 'bn_relu_fwd')
    torch.cuda.synchronize()
    close(y.cpu().numpy(), bn_relu(x, g, be)[0], 2e-5, 'relu(bn(x))')
