"""CPU: what pins the oracle (TensorFlow cannot run here -> parity unpinned by the reference):
known answers derivable from the reference source alone and fixtures
generated from the reference's one importable module (tests/golden/make_data_golden.py)."""
import os

import numpy as np
import pytest

import arch_and_hypers as A
from oracle import np_ops as O
from oracle.ref_net import RefNet

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'data_aug_golden.npz'))


def init_values(net, seed=0):
    rng = np.random.default_rng(seed)
    vals = {}
    for p in net._all_params:
        kind, scale = p.init
        vals[id(p)] = ((scale * rng.standard_normal(p.size)) if kind == 'normal' else
                       (np.ones(p.size) if kind == 'ones' else np.zeros(p.size))).reshape(p.shape)
    return vals


def test_ka1_ka2_op_counts():
    net = A.ac_chain(k_cpt=0.0)((32, 32, 3), (10,))
    blocks = [ℓ for ℓ in net.layers if ℓ.name == 'ReConvMax']
    assert [ℓ.n_ops for ℓ in blocks] == [1361664, 3907584, 2285568, 3833856, 2064384, 3538944, 1179648, 2359296]
    assert sum(ℓ.n_ops for ℓ in blocks) == 20530944
    assert [ℓ.n_ops for ℓ in net.layers if ℓ.name == 'LogReg'] == [2560, 2560, 5120, 5120, 10240, 10240, 20480, 20480]
    assert [ℓ.router.n_ops for ℓ in blocks if ℓ.router] == [4384, 4384, 8480, 8480, 16672, 16672, 33056]
    tr = [p for p in net._all_params if p.trainable]
    assert (len(tr), sum(p.size for p in tr)) == (178, 680798)
    assert sum(ℓ.n_ops for ℓ in A.sr_chain(8)((32, 32, 3), (10,)).layers) == 20551424
    assert sum(ℓ.n_ops for ℓ in A.sr_chain(8)((32, 32, 1), (10,)).layers) == 20551424 - (1361664 - 969984)
    assert len(list(net.layers)) == 17 and len(list(net.leaves)) == 8 and len(list(net.switches)) == 7
    # conv MACs of the oracle's own n_ops formula (layer_types.py:189-194)
    assert O.msconv_n_ops([(32, 32), (16, 16), (8, 8), (4, 4)], [(3, 3, 3, 16)] * 4, [(3, 3, 16, 16)] * 3) == 1361664


def test_ka3_ka4_ka5_ka6_initial_state():
    """Router outputs are exactly 0 at initialisation (arch_and_hypers.py:49): routing histogram
    [1,0,..], moc = 1 368 608, leaf p_tr = 2^-(j+1), uniform-softmax cross entropy ~ log 10 scale,
    TALR scale of leaf j ~ 2^(j+1)."""
    net = A.ac_chain(k_cpt=0.0)((32, 32, 3), (10,))
    ref = RefNet(net)
    ref.load_params(init_values(net))
    rng = np.random.default_rng(1)
    x0, y = rng.random((6, 32, 32, 3)), np.eye(10)[rng.integers(0, 10, 6)]
    res = ref.forward(x0, y, 'ev')
    st = ref.stats(res)
    assert np.array_equal(st['p_leaf'].mean(1), [1, 0, 0, 0, 0, 0, 0, 0])
    assert np.all(st['moc'] == 1361664 + 4384 + 2560)
    leaves = [ℓ for ℓ in net.layers if not ℓ.sinks]
    ptr = [float(res['out'][id(ℓ)]['p_tr'].mean()) for ℓ in leaves]
    assert abs(sum(ptr) - 1) < 1e-12
    for j, p in enumerate(ptr):
        assert abs(p - 2.0 ** -(min(j, 6) + 1)) < 1e-6
        assert abs(1 / np.sqrt(p * p) - 2.0 ** (min(j, 6) + 1)) < 1e-3 * 2.0 ** (j + 1)      # KA6
    # KA5: CrossEntropyError at a uniform softmax = log(n_cls)
    ce, _ = O.xent_eps(np.full((3, 10), 0.1), np.eye(10)[[0, 4, 9]])
    assert np.allclose(ce, np.log(10), atol=1e-6)


def test_ka7_and_reference_augmentation_fixtures():
    assert np.array_equal(GOLD['ka7'][:, :, 0], [[7.5, 0, 1, 2], [7.5, 4, 5, 6], [7.5, 8, 9, 10], [7.5, 12, 13, 14]])
    assert np.array_equal(O.shift_fill_mean(np.arange(16.0).reshape(4, 4, 1), 0, -1), GOLD['ka7'])
    from lib import data
    for k in range(3):
        seed, n, r = GOLD['case%d_args' % k]
        np.random.seed(int(seed))
        xb, yb = data.augmented_batch(GOLD['x0'], GOLD['y'], int(n), GOLD['m_sym'], int(r))
        assert xb.dtype == np.float64 and np.array_equal(xb, GOLD['case%d_x' % k])
        assert np.array_equal(yb, GOLD['case%d_y' % k])
    np.random.seed(5)
    xb, yb = data.batch(GOLD['x0'], GOLD['y'], 7)
    assert np.array_equal(xb, GOLD['batch_idx_x']) and np.array_equal(yb, GOLD['batch_idx_y'])


def test_schedules():
    assert A.λ_lrn(0) == 0.1 and abs(A.λ_lrn(10000) - 0.05) < 1e-15
    assert A.τ_ds(20000) == 0.5 and abs(A.τ_cr(20000) - 0.05) < 1e-15
    assert A.k_cpts == [0.0, 1e-9, 2e-9, 4e-9, 8e-9, 1.6e-8, 3.2e-8, 6.4e-8]
    assert (A.n_iter, A.t_log, A.batch_size) == (80000, 2500, 128)


def test_oracle_net_gradients_match_numpy_ops():
    """The whole-net oracle's autograd vs. the hand-derived NumPy backward for one block + head."""
    import torch
    net = A.sr_chain(1)((32, 32, 3), (10,))
    ref = RefNet(net)
    vals = init_values(net, 3)
    ref.load_params(vals)
    rng = np.random.default_rng(2)
    n = 3
    x0, y = rng.random((n, 32, 32, 3)), np.eye(10)[rng.integers(0, 10, n)]
    res = ref.train_step(x0, y, 0.0)
    blk = [ℓ for ℓ in net.layers if ℓ.name == 'ReConvMax'][0]
    conv, mbn, _ = blk.comps
    head = [ℓ for ℓ in net.layers if ℓ.name == 'LogReg'][0]
    P = lambda p: vals[id(p)]
    wh = [P(getattr(conv.params, 'w_horz_%i' % i)) for i in range(4)]
    wv = [P(getattr(conv.params, 'w_vert_%i' % i)) for i in range(3)]
    b = [P(getattr(conv.params, 'b_%i' % i)) for i in range(4)]
    xs = O.pyramid(x0, 4)
    s = O.msconv_fwd(xs, wh, wv, b)
    bn = [O.bn_train(s[i], P(mbn.comps[i].params.γ), P(mbn.comps[i].params.β)) for i in range(4)]
    a3 = O.relu(bn[3][0])
    lt = head.comps[1]
    z = O.lintrans(a3, P(lt.params.w), P(lt.params.b))
    p = O.softmax(z)
    gz = O.softmax_bwd(p, O.xent_eps_bwd(p, y, np.full(n, 1.0 / n)))
    ga3, gw, gb = O.lintrans_bwd(a3, P(lt.params.w), gz)
    G = lambda prm: res['grads'][id(prm)].numpy()
    k_l2 = lt.hypers.k_l2
    assert np.allclose(gw + 2 * k_l2 * P(lt.params.w), G(lt.params.w), atol=1e-12)
    g_out = [np.zeros_like(s[i]) for i in range(4)]
    g_out[3], dg, db = O.bn_train_bwd(s[3], P(mbn.comps[3].params.γ), bn[3][1], bn[3][2], O.relu_bwd(bn[3][0], ga3))
    assert np.allclose(dg, G(mbn.comps[3].params.γ), atol=1e-12)
    _, dwh, dwv, dbias, _ = O.msconv_bwd(xs, wh, wv, b, s, g_out)
    for i in range(4):
        assert np.allclose(dwh[i] + 2 * conv.hypers.k_l2 * wh[i], G(getattr(conv.params, 'w_horz_%i' % i)), atol=1e-11)
    for i in range(3):
        assert np.allclose(dwv[i] + 2 * conv.hypers.k_l2 * wv[i], G(getattr(conv.params, 'w_vert_%i' % i)), atol=1e-11)


@pytest.mark.parametrize('kind', ['actor', 'critic'])
def test_route_ref_agrees_with_whole_net_oracle(kind):
    """oracle/route_ref.py (the kernel-level oracle of mpnn_route, on bare tables) against the
    whole-net restatement oracle/ref_net.py on a shipped chain: same p_tr / p_ev / loss and the same
    gradients w.r.t. the router outputs and the leaf errors."""
    import torch
    import arch_and_hypers as A
    from oracle.ref_net import RefNet
    from oracle.route_ref import Tree, route
    mk = A.ac_chain if kind == 'actor' else A.cr_chain
    net = mk(k_cpt=8e-9)((32, 32, 3), (10,))
    rng = np.random.default_rng(3)
    vals = {}
    for p in net._all_params:
        k, scale = p.init
        vals[id(p)] = ((scale * rng.standard_normal(p.size)) if k == 'normal' else
                       (np.ones(p.size) if k == 'ones' else np.zeros(p.size))).reshape(p.shape)
    for ℓ in net.layers:
        if ℓ.router is not None:
            w = ℓ.router.comps[-1].params.w
            vals[id(w)] = rng.standard_normal(w.shape) * 0.5
    ref = RefNet(net)
    ref.load_params(vals)
    n = 6
    x0 = rng.random((n, 32, 32, 3))
    y = np.eye(10)[rng.integers(0, 10, n)]
    τ = 0.6
    res = ref.forward(x0, y, 'tr', τ=τ)
    R = lambda ℓ: res['out'][id(ℓ)]
    layers = list(net.layers)
    index = {id(ℓ): i for i, ℓ in enumerate(layers)}
    tree = Tree([dict(sinks=[index[id(s)] for s in ℓ.sinks]) for ℓ in layers])
    rs = [R(layers[i].router)['x'] for i in tree.switches]
    for t in rs:
        t.retain_grad()
    ce = [R(layers[i])['c_err'] for i in tree.leaves]
    for t in ce:
        t.retain_grad()
    res['c_tot'].backward()
    ops = [float(R(ℓ)['n_ops'] + (R(ℓ.router)['n_ops'] if ℓ.router is not None else 0)) for ℓ in layers]
    ϕ = net.hypers
    out = route(kind, tree, [t.detach().numpy() for t in rs], np.stack([t.detach().numpy() for t in ce]),
                np.stack([R(layers[i])['δ_cor'].numpy() for i in tree.leaves]), ops, τ=τ, ϵ=ϕ.ϵ, k_cpt=ϕ.k_cpt,
                k_dec=getattr(ϕ, 'k_dec', 0.0), k_cre=getattr(ϕ, 'k_cre', 0.0))
    for i, ℓ in enumerate(layers):
        assert np.allclose(out['p_tr'][i], R(ℓ)['p_tr'].detach().numpy(), rtol=1e-12, atol=1e-15)
        assert np.array_equal(out['p_ev'][i], R(ℓ)['p_ev'].numpy())
    for k, t in enumerate(rs):
        assert np.allclose(out['dr'][k], t.grad.numpy(), rtol=1e-9, atol=1e-15)
    for k, t in enumerate(ce):
        assert np.allclose(out['w_cerr'][k], t.grad.numpy(), rtol=1e-9, atol=1e-15)
