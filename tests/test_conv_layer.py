"""The single-scale `Conv` layer (scripts/lib/layer_types.py:55-74) through the C ABI
(mpnn_conv_nhwc_fwd / _dgrad / _wgrad, supp 3 and supp 1) against oracle/np_ops.conv_same*, and a
statically-routed net built from Conv / Rect / LinTrans (lib/_plan_conv.py) step by step against the
whole-net oracle, including the `res` identity initialisation and its L2 pull towards w_eq
(layer_types.py:46,52,65-72).

Tolerances: 2e-5 * (1 + max|ref|) forward / input gradients, 1e-4 * max|ref| weight gradients (fp32
sums over up to 128*32*32 pixels against float64); net steps as tests/test_net_parity.py (1e-4,
ReLU decisions read back from the device)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lib import _hip
from oracle import np_ops as O
from hiputil import DEV, dev, stream, pack_weights


def close(a, b, tol, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b).max()
    assert err <= tol * (1 + np.abs(b).max()), (what, err, np.abs(b).max())


CASES = [(3, 5, 16, 16, 3, 16, True), (3, 9, 8, 8, 16, 32, True), (3, 6, 4, 4, 64, 64, False), (3, 2, 32, 32, 16, 16, True),
         (1, 5, 16, 16, 3, 16, False), (1, 7, 8, 8, 16, 32, True), (1, 3, 4, 4, 64, 128, True), (1, 2, 32, 32, 20, 24, True),
         (1, 130, 4, 4, 128, 10, False)]


@pytest.mark.parametrize('supp,n,H,W,ci,co,relu', CASES)
def test_conv_nhwc_fwd_dgrad_wgrad(supp, n, H, W, ci, co, relu):
    lib = _hip.load()
    rng = np.random.default_rng(supp * 1000 + n + ci)
    x = rng.standard_normal((n, H, W, ci)).astype(np.float32)
    w = (rng.standard_normal((supp, supp, ci, co)) / np.sqrt(supp * supp * ci)).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)
    g = rng.standard_normal((n, H, W, co)).astype(np.float32)
    xa = np.maximum(x, 0) if relu else x                     # act(x): the Rect in front of this Conv
    xd, wd, bd, gd = dev(x), dev(w), dev(b), dev(g)
    mode = _hip.ACT_RELU if relu else _hip.ACT_IDENTITY
    if supp == 3:
        fw, bw = pack_weights([w])
    # forward
    out = torch.empty((n, H, W, co), device=DEV)
    a = _hip.ConvNhwcFwdArgs()
    a.a = _hip.act(xd, ci, mode)
    a.w = fw[0].data_ptr() if supp == 3 else wd.data_ptr()
    a.bias, a.out = bd.data_ptr(), out.data_ptr()
    a.n, a.H, a.W, a.Cout, a.supp = n, H, W, co, supp
    _hip.check(lib.mpnn_conv_nhwc_fwd(C.byref(a), stream()), 'conv_nhwc_fwd')
    torch.cuda.synchronize()
    close(out.cpu().numpy(), O.conv_same(xa, w) + b, 2e-5, 'forward')
    # gradients
    gx_ref, gw_ref = O.conv_same_bwd(xa, w, g)
    dw, db = torch.zeros_like(wd), torch.zeros(co, device=DEV)
    wg = _hip.ConvNhwcWgradArgs()
    wg.a, wg.g, wg.dw, wg.db = a.a, gd.data_ptr(), dw.data_ptr(), db.data_ptr()
    wg.n_split, wg.n, wg.H, wg.W, wg.Cout, wg.supp = 1, n, H, W, co, supp
    _hip.check(lib.mpnn_conv_nhwc_wgrad(C.byref(wg), stream()), 'conv_nhwc_wgrad')
    torch.cuda.synchronize()
    err = np.abs(dw.cpu().numpy() - gw_ref).max()
    assert err <= 1e-4 * np.abs(gw_ref).max(), ('dW', err)
    err = np.abs(db.cpu().numpy() - g.sum((0, 1, 2), dtype=np.float64)).max()
    assert err <= 1e-4 * np.abs(g.sum((0, 1, 2))).max() + 1e-5, ('db', err)
    if supp == 3 and ci % 16:
        return                                               # no backward pack: first-layer shapes have no input gradient
    dx = torch.empty((n, H, W, ci), device=DEV)
    scratch = torch.zeros(2 * max(ci, 16) * 16, device=DEV, dtype=torch.float64)
    d = _hip.ConvNhwcDgradArgs()
    d.g, d.Cg = gd.data_ptr(), co
    d.w = bw[0].data_ptr() if supp == 3 else wd.data_ptr()
    if relu:
        d.relu_src, d.scratch = xd.data_ptr(), scratch.data_ptr()
    d.dx = dx.data_ptr()
    d.n, d.H, d.W, d.Cin, d.supp = n, H, W, ci, supp
    _hip.check(lib.mpnn_conv_nhwc_dgrad(C.byref(d), stream()), 'conv_nhwc_dgrad')
    torch.cuda.synchronize()
    want = gx_ref * (x > 0) if relu else gx_ref
    close(dx.cpu().numpy(), want, 2e-5, 'input gradient')


def conv_net(res=False, k_l2=1e-4):
    from lib.layer_types import Chain, Conv, CrossEntropyError, LinTrans, Rect, Softmax
    from lib.net_types import SRNet

    def make_net(x0_shape, y_shape):
        head = Chain(name='LogReg', comps=[LinTrans(n_chan=y_shape[0], k_l2=k_l2), Softmax(), CrossEntropyError()])
        root = Chain(name='ConvNet', sinks=[head], comps=[
            Conv(n_chan=16, supp=3, k_l2=k_l2), Rect(),
            Conv(n_chan=16, supp=1, k_l2=k_l2, res=res), Rect(),
            Conv(n_chan=32, supp=3, k_l2=k_l2), Rect(),
            Conv(n_chan=32, supp=1, k_l2=k_l2, res=res)])          # (no Rect before the head)
        return SRNet(x0_shape=x0_shape, y_shape=y_shape, root=root)
    return make_net


@pytest.mark.parametrize('res,n', [(False, 24), (True, 24), (True, 5)])
def test_conv_net_training_steps_vs_oracle(res, n):
    """(n = 5: fewer pixel tiles than the weight-gradient split the engine sized for its capacity of 128 samples -- round 5
    found the slab reduction adding splits nobody had written, through the data-parallel test of this engine)"""
    from oracle.ref_net import RefNet
    net = conv_net(res)((16, 16, 3), (10,))
    eng = net.engine()
    assert type(eng).__name__ == 'ConvEngine'
    eng.init_params(4)
    if res:      # the identity part of the initialisation (layer_types.py:65-72)
        w = net.root.comps[2].params.w
        assert abs(float(w.numpy()[0, 0].diagonal().mean()) - 1) < 0.5 and w.eq is not None
    ref = RefNet(net)
    lr = 0.05
    for t in range(3):
        rng = np.random.default_rng(t)
        x0 = rng.random((n, 16, 16, 3)).astype(np.float32)
        y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, n)]
        ref.load_params()
        for p in net._all_params:
            ref.accum[id(p)] = torch.tensor(p.accum.cpu().numpy().astype(np.float64).reshape(p.shape))
        before = {id(p): p.data.clone() for p in net._all_params}
        net.train.run({net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: lr})
        # ReLU decisions as the device took them: the sign of the stored pre-activation maps
        forced, k = {}, 0
        for j, c in enumerate(net.root.comps):
            if type(c).__name__ == 'Rect':
                forced[('relu', id(net.root), j)] = (eng.out[k - 1][:n] > 0).cpu().numpy()
            else:
                k += 1
        res_ = ref.train_step(x0, y, lr, forced=forced)
        ce = res_['out'][id(net.root.sinks[0])]['c_err'].detach().numpy()
        assert np.abs(net.root.sinks[0].c_err.cpu().numpy() - ce).max() < 2e-4 * (1 + np.abs(ce).max())
        for p in net._all_params:
            v0 = before[id(p)].cpu().numpy().astype(np.float64)
            g_ref = res_['grads'][id(p)].numpy().reshape(-1)
            gp = p.grad.cpu().numpy().astype(np.float64)
            if p.l2:                                        # the L2 term lives in the optimizer kernel
                gp = gp + 2 * p.l2 * (v0 - (np.asarray(p.eq, np.float64).reshape(-1) if p.eq is not None else 0))
            assert np.abs(gp - g_ref).max() <= 1e-4 * np.abs(g_ref).max() + 1e-6, ('grad', p.owner.name, p.name, t)
            d = p.data.cpu().numpy().astype(np.float64) - v0
            d_ref = ref.V(p).detach().numpy().reshape(-1) - v0
            assert np.abs(d - d_ref).max() <= 1e-4 * np.abs(d_ref).max() + 1e-7, ('update', p.owner.name, p.name, t)
    # evaluation: any batch size through mpnn_exit_ev
    x0 = np.random.default_rng(9).random((200, 16, 16, 3)).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[np.random.default_rng(9).integers(0, 10, 200)]
    net.eval({net.x0: x0, net.y: y})
    ref.load_params()
    r = ref.forward(x0, y, 'ev')
    leaf = net.root.sinks[0]
    ce = r['out'][id(leaf)]['c_err'].detach().numpy()
    assert np.abs(leaf.c_err.cpu().numpy() - ce).max() < 2e-4 * (1 + np.abs(ce).max())
    assert np.array_equal(leaf.δ_cor.cpu().numpy(), r['out'][id(leaf)]['δ_cor'].numpy())
    st = net.state()
    assert abs(float(st[(net, 'acc')].mean()) - float(r['out'][id(leaf)]['δ_cor'].mean())) < 1e-6
    assert float(st[(net, 'moc')].mean()) == net.root.n_ops + leaf.n_ops


def test_lintrans_res_is_accepted_by_the_chain_engine():
    """LinTrans(res=True) (layer_types.py:46,52): its L2 term pulls towards the identity part w_eq."""
    import arch_and_hypers as A
    from lib.layer_types import Chain, CrossEntropyError, LinTrans, Select, Softmax
    from lib.net_types import SRNet
    head = Chain(name='LogReg', comps=[Select(i=-1), LinTrans(n_chan=10, k_l2=1e-2, res=True), Softmax(), CrossEntropyError()])
    net = SRNet(x0_shape=(32, 32, 3), y_shape=(10,), root=A.pyr(A.rcm(0, head)))
    eng = net.engine()
    eng.init_params(2)
    w = head.comps[1].params.w
    assert eng.w_eq is not None and w.eq is not None
    rng = np.random.default_rng(0)
    x0 = rng.random((8, 32, 32, 3)).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, 8)]
    v0 = w.numpy().reshape(-1).astype(np.float64)
    net.train.run({net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: 0.1, net.μ_lrn: 0.0})
    g = w.grad.cpu().numpy().astype(np.float64) + 2 * 1e-2 * (v0 - np.asarray(w.eq, np.float64).reshape(-1))
    d = w.numpy().reshape(-1).astype(np.float64) - v0
    assert np.abs(d + 0.1 * g).max() <= 1e-6 * np.abs(d).max() + 1e-9


@pytest.mark.parametrize('win,step', [(2, 2), (3, 2), (3, 1), (2, 3), (1, 1)])
def test_maxpool_kernels_against_numpy(win, step):
    """mpnn_maxpool_fwd / _bwd (layer_types.py:86-94) on a ragged 7 x 9 map with MANY ties (values quantised to
    eighths): output and first-maximum gradient routing exact against oracle/np_ops.max_pool_same."""
    lib = _hip.load()
    rng = np.random.default_rng(win * 10 + step)
    n, H, W, Cc = 5, 7, 9, 12
    x = (np.round(rng.standard_normal((n, H, W, Cc)) * 8) / 8).astype(np.float32)
    y_ref, arg = O.max_pool_same(x, win, step)
    dy = rng.standard_normal(y_ref.shape).astype(np.float32)
    dx_ref = O.max_pool_same_bwd(dy.astype(np.float64), arg, x.shape)
    xd, dyd = dev(x), dev(dy)
    yd = torch.empty(y_ref.shape, device=DEV)
    dxd = torch.empty(x.shape, device=DEV)
    st = stream()
    _hip.check(lib.mpnn_maxpool_fwd(xd.data_ptr(), yd.data_ptr(), None, n, H, W, Cc, win, step, 0, st), 'maxpool_fwd')
    _hip.check(lib.mpnn_maxpool_bwd(xd.data_ptr(), yd.data_ptr(), None, dyd.data_ptr(), dxd.data_ptr(), n, H, W, Cc, win, step, 0, st), 'maxpool_bwd')
    torch.cuda.synchronize()
    assert np.array_equal(yd.cpu().numpy(), y_ref)
    assert np.abs(dxd.cpu().numpy() - dx_ref).max() <= 1e-6 * (1 + np.abs(dx_ref).max())


def test_global_maxpool_kernels_against_numpy():
    """GlobalMaxPool (layer_types.py:96-100): tf.reduce_max over H x W; ties SHARE the gradient."""
    lib = _hip.load()
    rng = np.random.default_rng(3)
    n, H, W, Cc = 6, 8, 8, 20
    x = (np.round(rng.standard_normal((n, H, W, Cc)) * 4) / 4).astype(np.float32)
    y_ref, cnt_ref = O.global_max_pool(x)
    assert cnt_ref.max() > 1                       # the quantisation produced ties
    dy = rng.standard_normal(y_ref.shape).astype(np.float32)
    dx_ref = (x == y_ref[:, None, None, :]) * (dy / cnt_ref)[:, None, None, :]
    xd, dyd = dev(x), dev(dy)
    yd, cd, dxd = torch.empty((n, Cc), device=DEV), torch.empty((n, Cc), device=DEV), torch.empty(x.shape, device=DEV)
    st = stream()
    _hip.check(lib.mpnn_maxpool_fwd(xd.data_ptr(), yd.data_ptr(), cd.data_ptr(), n, H, W, Cc, 0, 0, 1, st), 'maxpool_fwd')
    _hip.check(lib.mpnn_maxpool_bwd(xd.data_ptr(), yd.data_ptr(), cd.data_ptr(), dyd.data_ptr(), dxd.data_ptr(), n, H, W, Cc, 0, 0, 1, st), 'maxpool_bwd')
    torch.cuda.synchronize()
    assert np.array_equal(yd.cpu().numpy(), y_ref) and np.array_equal(cd.cpu().numpy(), cnt_ref)
    assert np.abs(dxd.cpu().numpy() - dx_ref).max() <= 1e-6


def pooled_conv_net():
    from lib.layer_types import Chain, Conv, CrossEntropyError, GlobalMaxPool, LinTrans, MaxPool, Rect, Softmax
    from lib.net_types import SRNet

    def make_net(x0_shape, y_shape):
        head = Chain(name='LogReg', comps=[LinTrans(n_chan=y_shape[0], k_l2=1e-4), Softmax(), CrossEntropyError()])
        # (MaxPool hypers as the reference interprets them: window = stride, step = supp -- layer_types.py:90-94)
        root = Chain(name='ConvStack', sinks=[head], comps=[
            Conv(n_chan=16, supp=3, k_l2=1e-4), Rect(), MaxPool(stride=2, supp=2),
            Conv(n_chan=32, supp=3, k_l2=1e-4), MaxPool(stride=3, supp=2), Rect(),
            Conv(n_chan=32, supp=1, k_l2=1e-4), Rect(), GlobalMaxPool()])
        return SRNet(x0_shape=x0_shape, y_shape=y_shape, root=root)
    return make_net


def test_conv_net_with_pooling_layers_vs_oracle():
    """A Chain that mixes Conv / Rect / MaxPool / GlobalMaxPool (the free composition of the single-scale layer
    family, layer_types.py:55-100): 32x32 -> pool -> 16x16 -> overlapping pool -> 8x8 -> global.  Training steps
    against the float64 oracle with the device's ReLU decisions; the pooling arg-maxes are left to both sides (a
    near-tie that flips between fp32 and float64 would show as a large error: seeds are fixed)."""
    from oracle.ref_net import RefNet
    net = pooled_conv_net()((32, 32, 3), (10,))
    eng = net.engine()
    assert type(eng).__name__ == 'ConvEngine'
    assert [tuple(o.shape[1:]) for o in eng.out] == [(32, 32, 16), (16, 16, 16), (16, 16, 32), (8, 8, 32), (8, 8, 32), (1, 1, 32)]
    eng.init_params(6)
    ref = RefNet(net)
    n, lr = 16, 0.05
    for t in range(2):
        rng = np.random.default_rng(40 + t)
        x0 = rng.random((n, 32, 32, 3)).astype(np.float32)
        y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, n)]
        ref.load_params()
        for p in net._all_params:
            ref.accum[id(p)] = torch.tensor(p.accum.cpu().numpy().astype(np.float64).reshape(p.shape))
        before = {id(p): p.data.clone() for p in net._all_params}
        net.train.run({net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: lr})
        # ReLU decisions as the device took them: the sign of the stored map a Rect applies to
        forced, k = {}, 0
        for j, c in enumerate(net.root.comps):
            if type(c).__name__ == 'Rect':
                forced[('relu', id(net.root), j)] = (eng.out[k - 1][:n] > 0).cpu().numpy()
            else:
                k += 1
        res_ = ref.train_step(x0, y, lr, forced=forced)
        ce = res_['out'][id(net.root.sinks[0])]['c_err'].detach().numpy()
        assert np.abs(net.root.sinks[0].c_err.cpu().numpy() - ce).max() < 2e-4 * (1 + np.abs(ce).max())
        for p in net._all_params:
            v0 = before[id(p)].cpu().numpy().astype(np.float64)
            g_ref = res_['grads'][id(p)].numpy().reshape(-1)
            gp = p.grad.cpu().numpy().astype(np.float64)
            if p.l2:
                gp = gp + 2 * p.l2 * v0
            assert np.abs(gp - g_ref).max() <= 1e-4 * np.abs(g_ref).max() + 1e-6, ('grad', p.owner.name, p.name, t)
            d = p.data.cpu().numpy().astype(np.float64) - v0
            d_ref = ref.V(p).detach().numpy().reshape(-1) - v0
            assert np.abs(d - d_ref).max() <= 1e-4 * np.abs(d_ref).max() + 1e-7, ('update', p.owner.name, p.name, t)
    x0 = np.random.default_rng(9).random((70, 32, 32, 3)).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[np.random.default_rng(9).integers(0, 10, 70)]
    net.eval({net.x0: x0, net.y: y})
    ref.load_params()
    r = ref.forward(x0, y, 'ev')
    leaf = net.root.sinks[0]
    ce = r['out'][id(leaf)]['c_err'].detach().numpy()
    assert np.abs(leaf.c_err.cpu().numpy() - ce).max() < 2e-4 * (1 + np.abs(ce).max())
    assert float(net.state()[(net, 'moc')].mean()) == net.root.n_ops + leaf.n_ops


def test_conv_net_with_40_classes_runs_on_the_any_width_exit_kernels():
    """n_cls > 16 on the single-scale engine: the any-width head kernels (csrc/exit_gen.hip), a ReLU in front of the
    head masked through mpnn_bn_bwd_reduce.  Gradients of every parameter against the float64 oracle."""
    from oracle.ref_net import RefNet
    net = pooled_conv_net()((32, 32, 3), (40,))           # (ends ... Rect, GlobalMaxPool: the head's dX needs the mask)
    eng = net.engine()
    assert type(eng).__name__ == 'ConvEngine' and eng.generic and eng.stages[-1][1]
    eng.init_params(4)
    ref = RefNet(net)
    n, lr = 20, 0.05
    rng = np.random.default_rng(1)
    x0 = rng.random((n, 32, 32, 3)).astype(np.float32)
    y = np.eye(40, dtype=np.float32)[rng.integers(0, 40, n)]
    ref.load_params()
    before = {id(p): p.data.clone() for p in net._all_params}
    net.train.run({net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: lr})
    forced, k = {}, 0
    for j, c in enumerate(net.root.comps):
        if type(c).__name__ == 'Rect':
            forced[('relu', id(net.root), j)] = (eng.out[k - 1][:n] > 0).cpu().numpy()
        else:
            k += 1
    res_ = ref.train_step(x0, y, lr, forced=forced)
    for p in net._all_params:
        v0 = before[id(p)].cpu().numpy().astype(np.float64)
        g_ref = res_['grads'][id(p)].numpy().reshape(-1)
        gp = p.grad.cpu().numpy().astype(np.float64)
        if p.l2:
            gp = gp + 2 * p.l2 * (v0 - (np.asarray(p.eq, np.float64).reshape(-1) if p.eq is not None else 0))
        assert np.abs(gp - g_ref).max() <= 1e-4 * np.abs(g_ref).max() + 1e-6, ('grad', p.owner.name, p.name)
    net.eval({net.x0: x0, net.y: y})
    r = ref.forward(x0, y, 'ev')              # (the oracle still holds the pre-step parameters: reload)
    ref.load_params()
    r = ref.forward(x0, y, 'ev')
    leaf = net.root.sinks[0]
    ce = r['out'][id(leaf)]['c_err'].detach().numpy()
    assert np.abs(leaf.c_err.cpu().numpy() - ce).max() < 2e-4 * (1 + np.abs(ce).max())
