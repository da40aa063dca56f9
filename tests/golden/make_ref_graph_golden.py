"""Generates tests/golden/ref_graph_golden.npz by running the REFERENCE'S OWN graph-assembly code
(/root/reference/scripts/lib/{layer_types,net_types}.py and scripts/arch_and_hypers.py, imported
unmodified) on top of tests/golden/tf_standin.py, a float64 torch stand-in for the TensorFlow calls
they make.  Run in the build container (the reference tree does not travel):

    python tests/golden/make_ref_graph_golden.py

What the vectors pin and what they do not: see the header of tf_standin.py -- the reference's Python
(scale selection, routing probabilities, epsilon floors, hard routing, critic costs, stop-gradients,
cost assembly, TALR scales, the k_cpt column, Momentum wiring) YES; TensorFlow's operator semantics
NO (the stand-in implements them from the same assumptions as oracle/np_ops.py).

Inputs are not stored: tests/test_ref_graph_golden.py regenerates weights and batches from the same
seeded legacy numpy streams (`case_inputs`, `param_value` below are imported by the test).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/scripts'

# (temperatures: moderate on purpose.  A peaked softmax puts p_tr of the deep nodes at the epsilon floor,
# the TALR scale 1/sqrt(mean p_tr^2) then reaches 1e3..1e6 and multiplies the fp32 rounding noise of
# analytically-zero gradients -- conv biases ahead of BatchNorm -- into visible parameter changes: a property
# of the update rule in fp32, not something a float64 fixture can hold the kernels to.)
CASES = {
    'ac': dict(ctor='ac_chain', hypers=dict(k_cpt=1.6e-8), tau=0.7, n=4),
    'ac_notalr_nokdec': dict(ctor='ac_chain', hypers=dict(k_cpt=4e-9, talr=False, k_dec=0.0), tau=1.0, n=3),
    'ac_dyn': dict(ctor='ac_chain', hypers=dict(dyn_k_cpt=True), tau=0.8, n=4, dyn=True),
    'cr': dict(ctor='cr_chain', hypers=dict(k_cpt=8e-9), tau=0.5, n=4),
    'cr_opt_cls': dict(ctor='cr_chain', hypers=dict(k_cpt=8e-9, optimistic=True, use_cls_err=True), tau=0.3, n=3),
    'sr3': dict(ctor='sr_chain', args=(3,), hypers={}, tau=None, n=3),
    # the deepest statically-routed chain (scripts/train-nets:81-88), on CIFAR-shaped and on MNIST-shaped (one channel,
    # prep-data:35-38) inputs.  (Seeds: `seed_of` below.)
    'sr8': dict(ctor='sr_chain', args=(8,), hypers={}, tau=None, n=3),
    'sr8_mnist': dict(ctor='sr_chain', args=(8,), hypers={}, tau=None, n=3, c0=1),
    # a 3-way switch over two sub-chains, built from the spec's own rcm / reg / pyr (the reference's dr_tree
    # cannot run at this revision: `y_shape` scoping bug, arch_and_hypers.py:99-106)
    'ac_tree3': dict(ctor='small_tree', net='ActorNet', hypers=dict(k_cpt=1.6e-8), tau=0.8, n=4),
    'cr_tree3': dict(ctor='small_tree', net='CriticNet', hypers=dict(k_cpt=8e-9, optimistic=True), tau=0.4, n=4),
    # the router factor alpha_rtr * lr_scale (net_types.py:25-33) away from 1 -- WITHOUT TALR (lr_scale = 1: the factor
    # must still apply; rounds 1-4 dropped it in the oracle and in the kernel alike, and no fixture had alpha_rtr != 1),
    # with TALR, and on a critic net without TALR
    'ac_notalr_artr2': dict(ctor='ac_chain', hypers=dict(k_cpt=4e-9, talr=False, α_rtr=2.0), tau=1.0, n=3, seed=10),
    'cr_artr3': dict(ctor='cr_chain', hypers=dict(k_cpt=8e-9, α_rtr=3.0), tau=0.5, n=3, seed=11),
    'cr_notalr_artr': dict(ctor='cr_chain', hypers=dict(k_cpt=2e-9, talr=False, α_rtr=0.5, k_cre=0.01), tau=0.2, n=3, seed=12),
    # LAYER-level keyword arguments (round 6; `layer_kwargs_chain` below): MultiscaleBatchNorm(d=0.5, ϵ=1e-3) on block 1
    # -- which the reference accepts and DISCARDS (layer_types.py:246; rounds 1-5 forwarded them: 75 of 108 tensors of
    # this very net differed from the reference's code) --, the two router BatchNorms with hypers of their own (and
    # different from each other), CrossEntropyError(ϵ), per-layer k_l2 and a LinTrans(res=True) exit
    'ac_layer_kwargs': dict(ctor='layer_kwargs_chain', net='ActorNet', hypers=dict(k_cpt=1.6e-8), tau=0.7, n=4, seed=13),
    'cr_layer_kwargs': dict(ctor='layer_kwargs_chain', net='CriticNet', hypers=dict(k_cpt=8e-9), tau=0.5, n=4, seed=14),
}
# seeds of the first ten cases: their index in the sorted key list of the round they were generated in
_LEGACY = ['ac', 'ac_dyn', 'ac_notalr_nokdec', 'ac_tree3', 'cr', 'cr_opt_cls', 'cr_tree3', 'sr3', 'sr8', 'sr8_mnist']


def seed_of(key):
    return CASES[key].get('seed', _LEGACY.index(key) if key in _LEGACY else None)


def small_tree(A, NT, case):
    def make_net(x0_shape, y_shape):
        nc = y_shape[0]
        root = A.pyr(A.rcm(0, A.reg(nc),
                           A.rcm(1, A.reg(nc), A.rcm(2, A.reg(nc))),
                           A.rcm(1, A.reg(nc), A.rcm(2, A.reg(nc), A.rcm(3, A.reg(nc))))))
        return getattr(NT, case['net'])(x0_shape=x0_shape, y_shape=y_shape, root=root, **case['hypers'])
    return make_net


def layer_kwargs_chain(A, NT, case):
    """A 3-block chain built through the LAYER CLASSES of the side that runs (the module A imported them from), with
    keyword arguments the shipped spec never passes."""
    L = sys.modules[A.Chain.__module__]

    def router(bn1, bn2, k):
        return L.Chain(name='Router', comps=[
            L.Select(i=-1), L.LinTrans(n_chan=A.router_n_chan, k_l2=k, σ_w=1),
            L.BatchNorm(**bn1), L.Rect(), L.LinTrans(n_chan=A.router_n_chan, k_l2=3 * k, σ_w=1),
            L.BatchNorm(**bn2), L.Rect(), L.LinTrans(n_chan=2, k_l2=A.k_l2, σ_w=0)])

    def reg(nc, **kw):
        return L.Chain(name='LogReg', comps=[L.Select(i=-1), L.LinTrans(n_chan=nc, k_l2=A.k_l2, σ_w=1, **kw.get('lin', {})),
                                             L.Softmax(), L.CrossEntropyError(**kw.get('ce', {}))])

    def rcm(i, sinks, msbn, rt, k_l2=A.k_l2):
        return L.Chain(name='ReConvMax', sinks=sinks, router=rt, comps=[
            L.MultiscaleConvMax(n_chan=A.arch[i], supp=A.conv_supp, k_l2=k_l2, σ_w=1),
            L.MultiscaleBatchNorm(**msbn), L.MultiscaleRect()])

    def make_net(x0_shape, y_shape):
        nc = y_shape[0]
        b2 = rcm(2, (reg(nc, ce=dict(ϵ=1e-3)),), {}, None)
        b1 = rcm(1, (reg(nc, lin=dict(res=True)), b2), dict(d=0.5, ϵ=1e-3), router(dict(d=0.8, ϵ=1e-3), dict(d=0.6, ϵ=1e-4), 2e-4),
                 k_l2=5e-4)
        b0 = rcm(0, (reg(nc), b1), {}, router({}, dict(ϵ=1e-2), A.k_l2))
        root = L.Chain(name='ToPyramid', sinks=(b0,), router=None, comps=[L.ToPyramid(n_scales=len(A.arch[0]))])
        return getattr(NT, case['net'])(x0_shape=x0_shape, y_shape=y_shape, root=root, **case['hypers'])
    return make_net


def make_case(A, NT, case):
    if case['ctor'] == 'small_tree':
        return small_tree(A, NT, case)
    if case['ctor'] == 'layer_kwargs_chain':
        return layer_kwargs_chain(A, NT, case)
    return getattr(A, case['ctor'])(*case.get('args', ()), **case['hypers'])
K_CPTS = [0.0, 1e-9, 2e-9, 4e-9, 8e-9, 1.6e-8, 3.2e-8, 6.4e-8]
LR = 0.05


def param_value(name, shape, rng):
    """Injected parameter values (both sides draw them in the same order from the same stream)."""
    shape = tuple(shape)
    if name.startswith('w'):
        fan = max(1, int(np.prod(shape[:-1])))
        return rng.standard_normal(shape) * (1.5 / np.sqrt(fan))
    if name.startswith('b'):
        return rng.standard_normal(shape) * 0.1
    if name == 'γ':
        return 1 + 0.1 * rng.standard_normal(shape)
    if name == 'β':
        return 0.1 * rng.standard_normal(shape)
    if name == 'm_avg':
        return 0.1 * rng.standard_normal(shape)
    if name == 'v_avg':
        return 1 + 0.3 * rng.random_sample(shape)
    raise KeyError(name)


def case_inputs(case, seed):
    rng = np.random.RandomState(1000 + seed)
    n = case['n']
    x0 = rng.random_sample((n, 32, 32, case.get('c0', 3)))
    y = np.eye(10)[rng.randint(0, 10, n)]
    kc = np.asarray(K_CPTS)[rng.randint(0, len(K_CPTS), n)] if case.get('dyn') else None
    return x0, y, kc


def ordered_params(net, params_list_rec):
    """[(name, variable)] in tree order: every layer's own parameters, then its router's."""
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


def pad_r(v, width=3):
    v = np.asarray(v, np.float64)
    return np.concatenate([v, np.zeros((v.shape[0], width - v.shape[1]))], 1)


def digest(v):
    v = np.asarray(v, np.float64).reshape(-1)
    pos = [0, len(v) // 3, len(v) - 1]
    return [v.sum(), np.abs(v).sum(), v[pos[0]], v[pos[1]], v[pos[2]]]


def main():
    sys.path.insert(0, HERE)
    import tf_standin
    sys.modules['tensorflow'] = tf_standin
    sys.path.insert(0, REF)
    import lib.net_types as NT                       # the REFERENCE's modules
    import arch_and_hypers as A
    assert NT.__file__.startswith(REF) and A.__file__.startswith(REF)
    out = {}
    for key, case in sorted(CASES.items()):
        seed = seed_of(key)
        tf_standin.reset()
        net = make_case(A, NT, case)((32, 32, case.get('c0', 3)), (10,))
        rng = np.random.RandomState(seed)
        params = ordered_params(net, NT.params_list_rec)
        for name, var in params:
            var.load(param_value(name, var.data.shape, rng))
        x0, y, kc = case_inputs(case, seed)
        layers = list(net.layers)
        leaves = [ℓ for ℓ in layers if len(ℓ.sinks) == 0]
        switches = [ℓ for ℓ in layers if len(ℓ.sinks) > 1]
        feed = {net.x0: x0, net.y: y}
        if case['tau'] is not None:
            feed[net.τ] = case['tau']
        if kc is not None:
            feed[net.k_cpt] = kc
        has_ptr = hasattr(layers[0], 'p_tr')
        fetch = {'p_ev': [ℓ.p_ev for ℓ in layers], 'c_err': [ℓ.c_err for ℓ in leaves], 'd_cor': [ℓ.δ_cor for ℓ in leaves]}
        if has_ptr:
            fetch['p_tr'] = [ℓ.p_tr for ℓ in layers]
            fetch['r'] = [ℓ.router.x for ℓ in switches]
        for mode in ('ev', 'tr'):
            f = dict(feed)
            f[net.mode] = mode
            if mode == 'tr':
                # (forward values of the training graph, BEFORE the update: a separate run of the same feed
                # would move the moving averages, so snapshot and restore the non-trainable state)
                snap = [(v, v.data.detach().clone()) for _, v in params]
            for k, nodes in fetch.items():
                vals = tf_standin.run(nodes, f)
                if k == 'r':                       # switches have 2 or 3 sinks: zero-padded to the widest
                    vals = [pad_r(v) for v in vals]
                out['%s/%s/%s' % (key, mode, k)] = np.stack([np.broadcast_to(np.asarray(v, np.float64), np.asarray(vals[0]).shape)
                                                            if k != 'r' else np.asarray(v, np.float64) for v in vals])
            if mode == 'tr':
                for v, d in snap:
                    v.load(d.numpy())
        # one training step (momentum 0.9 from zero accumulators), then the state of every variable
        f = dict(feed)
        f[net.mode] = 'tr'
        f[net.λ_lrn] = LR
        net.train.run(f)
        out['%s/after' % key] = np.array([digest(v.data.detach().numpy()) for _, v in params])
        out['%s/names' % key] = np.array([n for n, _ in params])
        print(key, 'ok:', len(params), 'variables,', len(layers), 'nodes')
    np.savez_compressed(os.path.join(HERE, 'ref_graph_golden.npz'), **out)


if __name__ == '__main__':
    main()
