"""CPU: oracle/ref_net.py against vectors produced by the REFERENCE'S OWN graph-assembly code.

tests/golden/ref_graph_golden.npz was generated (tests/golden/make_ref_graph_golden.py, in the build
container) by importing the reference's scripts/lib/layer_types.py, scripts/lib/net_types.py and
scripts/arch_and_hypers.py UNMODIFIED on top of tests/golden/tf_standin.py, a float64 torch stand-in
for the TensorFlow calls they make.  The same seeded weights and batches go through the oracle here.

This pins the restatement against the reference's Python -- negative-index scale selection, parameter
naming and order, pi_tr with its epsilon floors, pi_ev, c_ev / c_opt / c_cre, every stop_gradient, the
cost assembly of all three net types, the TALR scales and the router factor, the k_cpt column, the
Momentum wiring -- to float64 rounding.  It does NOT pin TensorFlow's operator semantics (the stand-in
implements conv2d / max_pool / moments / argmax / resize / Momentum from the same assumptions as
oracle/np_ops.py): parity with the TensorFlow reference stays "unpinned" (DESIGN.md §4).
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import make_ref_graph_golden as M

GOLD = np.load(os.path.join(HERE, 'golden', 'ref_graph_golden.npz'))


def ordered(net):
    out = []
    for ℓ in net.layers:
        for scope in (ℓ, ℓ.router):
            if scope is None:
                continue

            def walk(l):
                for k, v in vars(l.params).items():
                    out.append((k, v))
                for c in getattr(l, 'comps', []):
                    walk(c)
            walk(scope)
    return out


@pytest.mark.parametrize('key', sorted(M.CASES))
def test_oracle_matches_the_reference_graph_code(key):
    import arch_and_hypers as A
    import lib.net_types as NT
    from oracle.ref_net import RefNet
    case = M.CASES[key]
    seed = M.seed_of(key)
    net = M.make_case(A, NT, case)((32, 32, case.get('c0', 3)), (10,))
    params = ordered(net)
    assert [n for n, _ in params] == list(GOLD['%s/names' % key])        # same parameters, same order
    rng = np.random.RandomState(seed)
    vals = {id(p): M.param_value(n, p.shape, rng) for n, p in params}
    ref = RefNet(net)
    ref.load_params(vals)
    x0, y, kc = M.case_inputs(case, seed)
    kw = {}
    if case['tau'] is not None:
        kw['τ'] = case['tau']
    if kc is not None:
        kw['k_cpt'] = kc
    layers = list(net.layers)
    leaves = [ℓ for ℓ in layers if not ℓ.sinks]
    switches = [ℓ for ℓ in layers if len(ℓ.sinks) > 1]

    def close(a, b, what):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        assert a.shape == b.shape, (what, a.shape, b.shape)
        assert np.abs(a - b).max() <= 1e-9 * (1 + np.abs(b).max()), (key, what, np.abs(a - b).max())
    for mode in ('ev', 'tr'):
        res = ref.forward(x0, y, mode, **kw)
        R = lambda ℓ: res['out'][id(ℓ)]
        n = len(x0)
        vec = lambda v: np.broadcast_to(v.detach().numpy() if hasattr(v, 'detach') else np.asarray(v, np.float64), (n,))
        assert np.array_equal(np.stack([vec(R(ℓ)['p_ev']) for ℓ in layers]), GOLD['%s/%s/p_ev' % (key, mode)])
        close(np.stack([vec(R(ℓ)['c_err']) for ℓ in leaves]), GOLD['%s/%s/c_err' % (key, mode)], mode + ' c_err')
        assert np.array_equal(np.stack([vec(R(ℓ)['δ_cor']) for ℓ in leaves]), GOLD['%s/%s/d_cor' % (key, mode)])
        if '%s/%s/p_tr' % (key, mode) in GOLD:
            close(np.stack([vec(R(ℓ)['p_tr']) for ℓ in layers]), GOLD['%s/%s/p_tr' % (key, mode)], mode + ' p_tr')
            close(np.stack([M.pad_r(R(ℓ.router)['x'].detach().numpy()) for ℓ in switches]), GOLD['%s/%s/r' % (key, mode)], mode + ' router.x')
    # one training step: every variable (parameters after TALR-scaled momentum update, moving averages)
    ref.train_step(x0, y, M.LR, **kw)
    after = np.array([M.digest(ref.V(p).detach().numpy()) for _, p in params])
    gold = GOLD['%s/after' % key]
    err = np.abs(after - gold) / (1e-12 + np.abs(gold).max(0, keepdims=True))
    assert err.max() <= 1e-9, (key, [params[i][0] for i in np.argwhere(err > 1e-9)[:, 0][:5]], err.max())


@pytest.mark.gpu
@pytest.mark.parametrize('key', sorted(M.CASES))
def test_product_matches_the_reference_graph_code(key):
    """The HIP path against the same vectors (fp32 kernels vs the float64 stand-in run of the reference's
    code): p_ev / delta_cor exact, per-sample values 2e-4, every variable after one training step within
    5e-4 of its scale (updates are ~1e-2 of the values, so this holds the update to a few percent; the
    TALR scale 1/sqrt(mean p_tr^2) of the deep nodes multiplies fp32 rounding into the update -- most
    visibly for the conv biases ahead of BatchNorm, whose true gradient is exactly zero: 2e-3 for those;
    the unused BatchNorms keep their moving averages)."""
    import torch
    import arch_and_hypers as A
    import lib.net_types as NT
    case = M.CASES[key]
    seed = M.seed_of(key)
    net = M.make_case(A, NT, case)((32, 32, case.get('c0', 3)), (10,))
    net.engine()
    params = ordered(net)
    rng = np.random.RandomState(seed)
    for n_, p in params:
        p.assign(M.param_value(n_, p.shape, rng))
    x0, y, kc = M.case_inputs(case, seed)
    feed = {net.x0: x0.astype(np.float32), net.y: y.astype(np.float32)}
    if case['tau'] is not None:
        feed[net.τ] = case['tau']
    if kc is not None:
        feed[net.k_cpt] = kc.astype(np.float32)
    layers = list(net.layers)
    leaves = [ℓ for ℓ in layers if not ℓ.sinks]
    switches = [ℓ for ℓ in layers if len(ℓ.sinks) > 1]

    def check(mode):
        g = lambda k: GOLD['%s/%s/%s' % (key, mode, k)]
        assert np.array_equal(np.stack([ℓ.p_ev.cpu().numpy() for ℓ in layers]), g('p_ev'))
        assert np.array_equal(np.stack([ℓ.δ_cor.cpu().numpy() for ℓ in leaves]), g('d_cor'))
        ce = np.stack([ℓ.c_err.cpu().numpy() for ℓ in leaves])
        assert np.abs(ce - g('c_err')).max() <= 2e-4 * (1 + np.abs(g('c_err')).max())
        if '%s/%s/p_tr' % (key, mode) in GOLD:
            assert np.abs(np.stack([ℓ.p_tr.cpu().numpy() for ℓ in layers]) - g('p_tr')).max() <= 2e-4
            r = np.stack([M.pad_r(ℓ.router.x.cpu().numpy()) for ℓ in switches])
            assert np.abs(r - g('r')).max() <= 2e-4 * (1 + np.abs(g('r')).max())
    net.eval(feed)
    check('ev')
    before = np.array([M.digest(p.numpy()) for _, p in params])
    net.train.run({**feed, net.mode: 'tr', net.λ_lrn: M.LR})
    torch.cuda.synchronize()
    check('tr')
    after = np.array([M.digest(p.numpy()) for _, p in params])
    gold = GOLD['%s/after' % key]
    scale = np.abs(gold).max(1, keepdims=True)
    err = (np.abs(after - gold) / (1e-12 + scale)).max(1)
    zero_grad = np.array([n_.startswith('b_') for n_, _ in params])         # MultiscaleConvMax biases: d/db == 0 through BatchNorm
    tol = np.where(zero_grad, 2e-3, 5e-4)
    assert (err <= tol).all(), (key, [(params[i][0], float(err[i])) for i in np.argwhere(err > tol)[:, 0][:5]])
    # The UPDATE itself (after - before; ~1e-2 of the values, so the check above holds it to a few percent
    # only): the linear digest entries (sum, three samples) of every trainable tensor's update against the
    # reference-graph vectors, 2e-3 of the update's scale.  The golden "before" is the injected float64 value,
    # the device's its fp32 rounding; each side subtracts its own.
    rng = np.random.RandomState(seed)
    before64 = np.array([M.digest(M.param_value(n_, p.shape, rng)) for n_, p in params])
    lin = [0, 2, 3, 4]
    upd_dev, upd_ref = (after - before)[:, lin], (gold - before64)[:, lin]
    trainable = np.array([p.trainable for _, p in params])
    # per-tensor scale: the update's largest linear-digest entry, floored by what fp32 rounding of the VALUE
    # allows (|sum| of a tensor of thousands of elements carries ~1e-7 of sum|v|)
    floor = 3e-7 * np.abs(gold[:, 1:2])
    uscale = np.abs(upd_ref).max(1, keepdims=True)
    uerr = (np.abs(upd_dev - upd_ref) - floor).max(1) / (1e-30 + uscale[:, 0])
    chk = trainable & ~zero_grad & (uscale[:, 0] > 0)
    worst = np.argsort(-np.where(chk, uerr, -1))[:5]
    print('%s: worst update errors (of the update scale): %s' % (key, [(params[i][0], float(uerr[i])) for i in worst]))
    over = np.argwhere(chk & (uerr > 2e-3))[:, 0]
    if len(over):
        # The vectors come from a FREE float64 run: where the fp32 device decided a max-pool near-tie or a ReLU
        # sign differently, the tensors downstream of that element legitimately differ.  Count those decisions
        # (device vs the oracle's free run, which reproduces the vectors to 1e-9): without one, nothing may exceed.
        from oracle.ref_net import RefNet
        from test_net_parity import count_flips
        ref = RefNet(net)
        rng = np.random.RandomState(seed)
        vals = {id(p): M.param_value(n_, p.shape, rng) for n_, p in params}
        ref.load_params(vals)
        kw = {}
        if case['tau'] is not None:
            kw['τ'] = case['tau']
        if kc is not None:
            kw['k_cpt'] = kc
        free = ref.forward(x0, y, 'tr', **kw)
        before_t = {id(p): torch.tensor(np.asarray(vals[id(p)], np.float32)) for _, p in params}
        flips, decisions = count_flips(net.engine(), free, len(x0), before_t)
        print('%s: %d of %d decisions differ from the free float64 run; %d tensors beyond 2e-3' % (key, flips, decisions, len(over)))
        assert 0 < flips <= max(3, 1e-4 * decisions) and len(over) <= 4 * flips and uerr[over].max() <= 0.1, \
            (key, flips, [(params[i][0], float(uerr[i])) for i in over[:8]])
