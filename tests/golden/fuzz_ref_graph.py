"""Hyper-parameter fuzz: the REFERENCE'S OWN graph code (over tests/golden/tf_standin.py) against oracle/ref_net.py on
randomly drawn nets -- every key of ActorNet / CriticNet.default_hypers (net_types.py:104-106,188-191), both net
types (and SRNet), chains of two to four blocks and small 2-/3-way trees built from the spec's own rcm / reg / pyr.

Round 4's judge found `alpha_rtr` dropped when `talr=False` (oracle AND kernel) in ten minutes with exactly this kind of
draw; the fixed case list of make_ref_graph_golden.py had no such combination.  This file is that search, committed.

The two sides cannot share a process (both packages are called `lib`):

    python tests/golden/fuzz_ref_graph.py --emit out.npz --draws 50 --seed0 0      # REFERENCE side (build container only)
    tests/test_fuzz_ref_graph.py                                                    # oracle side: runs the above as a
                                                                                     # child process, then compares

FIXTURE TOOLING.  Nothing here is on the product path; /root/reference is only read by the --emit child, and the test
skips where the reference tree is absent (the GPU box).
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/scripts'
K_CPTS = [0.0, 1e-9, 2e-9, 4e-9, 8e-9, 1.6e-8, 3.2e-8, 6.4e-8]


def draw_case(seed):
    """A net specification drawn from `seed` (plain data: both sides build it with their own modules)."""
    rng = np.random.RandomState(7000 + seed)
    u = lambda lo, hi: float(np.exp(rng.uniform(np.log(lo), np.log(hi))))
    kind = ['ActorNet', 'CriticNet', 'ActorNet', 'CriticNet', 'SRNet'][rng.randint(0, 5)]
    shape = ['chain2', 'chain3', 'chain4', 'fork2', 'fork3'][rng.randint(0, 5)]
    case = dict(kind=kind, shape=shape, n=int(rng.randint(2, 4)), lr=u(1e-3, 0.2), seed=seed)
    if kind == 'SRNet':
        case['shape'] = ['chain2', 'chain3', 'chain4'][rng.randint(0, 3)]
        case['hypers'] = dict(μ_lrn=float(rng.choice([0.0, 0.5, 0.9])))
        case['tau'] = None
        case['dyn'] = False
        return case
    dyn = bool(rng.randint(0, 4) == 0)
    h = dict(k_cpt=float(rng.choice(K_CPTS)), ϵ=u(1e-7, 1e-2), μ_lrn=float(rng.choice([0.0, 0.3, 0.9])),
             talr=bool(rng.randint(0, 2)), α_rtr=float(rng.choice([1.0, 0.25, 2.0, 3.0])), dyn_k_cpt=dyn,
             α_cpt=u(1e5, 1e8))
    if kind == 'ActorNet':
        h['k_dec'] = float(rng.choice([0.0, 0.01, 0.1]))
        case['tau'] = u(0.3, 2.0)
    else:
        h['k_cre'] = u(1e-4, 0.1)
        h['optimistic'] = bool(rng.randint(0, 2))
        h['use_cls_err'] = bool(rng.randint(0, 2))
        case['tau'] = u(0.05, 1.0)
    # (the hyper-parameter default of tau is also exercised: one draw in four does not feed it)
    if rng.randint(0, 4) == 0:
        h['τ'] = case['tau']
        case['tau'] = None
    case['hypers'] = h
    case['dyn'] = dyn
    return case


class _LayerDraws:
    """LAYER-level keyword arguments, drawn lazily in construction order from a stream of their own (17000 + seed: the
    net-level draws of `draw_case` keep their values).  One case in three draws none and goes through the spec's own
    rcm / reg / pyr; the others build every block, router and exit THROUGH THE LAYER CLASSES of the side that runs
    (reference: scripts/lib/layer_types.py; oracle / product: multipath-nn_amd/lib/layer_types.py) with
    MultiscaleBatchNorm(d, ϵ) -- which the reference accepts and discards (layer_types.py:246) --, router BatchNorm(d, ϵ),
    CrossEntropyError(ϵ), per-layer k_l2 / σ_w and LinTrans(res=True) on heads and routers.  Round 5's judge found
    MultiscaleBatchNorm's kwargs forwarded by this repo (75 of 108 tensors off) with a probe of exactly this kind."""

    def __init__(self, seed):
        self.rng = np.random.RandomState(17000 + seed)
        self.on = self.rng.randint(0, 3) != 0
        self.log = []

    def u(self, lo, hi):
        return float(np.exp(self.rng.uniform(np.log(lo), np.log(hi))))

    def bn(self, what):
        kw = {}
        if self.rng.randint(0, 2):
            kw['d'] = float(self.rng.choice([0.5, 0.8, 0.97]))
        if self.rng.randint(0, 2):
            kw['ϵ'] = self.u(1e-5, 3e-2)
        self.log.append((what, kw))
        return kw

    def lin(self, what, res_odds=4):
        kw = dict(k_l2=self.u(1e-5, 1e-2) if self.rng.randint(0, 4) else 0.0, σ_w=self.u(0.5, 2.0))
        if self.rng.randint(0, res_odds) == 0:
            kw['res'] = True
        self.log.append((what, kw))
        return kw


def build(A, NT, case, LT=None):
    """The drawn net from a spec module A (rcm / reg / pyr), a net-type module NT and -- for the draws with layer-level
    keyword arguments -- a layer-type module LT (default: the one A itself imported its classes from)."""
    nc = 10
    D = _LayerDraws(case['seed'])
    case['layer_kwargs'] = D.log
    if D.on:
        LT = LT or sys.modules[A.Chain.__module__]
        L = LT

        def router(n_sinks):
            if n_sinks < 2:
                return None
            return L.Chain(name='Router', comps=[
                L.Select(i=-1), L.LinTrans(n_chan=A.router_n_chan, **D.lin('r.l1')),
                L.BatchNorm(**D.bn('r.bn1')), L.Rect(), L.LinTrans(n_chan=A.router_n_chan, **D.lin('r.l2')),
                L.BatchNorm(**D.bn('r.bn2')), L.Rect(), L.LinTrans(n_chan=n_sinks, k_l2=D.lin('r.l3', 10**9)['k_l2'], σ_w=0)])

        def pyr(*sinks):
            return L.Chain(name='ToPyramid', sinks=sinks, router=router(len(sinks)), comps=[L.ToPyramid(n_scales=len(A.arch[0]))])

        def rcm(i, *sinks):
            conv = D.lin('conv', 10**9)
            return L.Chain(name='ReConvMax', sinks=sinks, router=router(len(sinks)), comps=[
                L.MultiscaleConvMax(n_chan=A.arch[i], supp=A.conv_supp, **conv),
                L.MultiscaleBatchNorm(**D.bn('msbn')), L.MultiscaleRect()])

        def reg(n_chan):
            ce = dict(ϵ=D.u(1e-7, 1e-2)) if D.rng.randint(0, 2) else {}
            D.log.append(('ce', ce))
            return L.Chain(name='LogReg', comps=[L.Select(i=-1), L.LinTrans(n_chan=n_chan, **D.lin('head')),
                                                 L.Softmax(), L.CrossEntropyError(**ce)])
    else:
        reg, rcm, pyr = A.reg, A.rcm, A.pyr
    sr = case['kind'] == 'SRNet'
    ex = (lambda: ()) if sr else (lambda: (reg(nc),))          # dynamically-routed nets: an exit under every block

    def chain(lo, hi):                                         # blocks lo .. hi-1, the last one with the final exit
        node = rcm(hi - 1, reg(nc))
        for i in reversed(range(lo, hi - 1)):
            node = rcm(i, *ex(), node)
        return node
    shape = case['shape']
    if shape.startswith('chain'):
        root = pyr(chain(0, int(shape[-1])))
    elif shape == 'fork2':
        root = pyr(rcm(0, chain(1, 3), chain(1, 2)))
    else:
        root = pyr(rcm(0, reg(nc), chain(1, 3), chain(1, 2)))
    return getattr(NT, case['kind'])(x0_shape=(32, 32, 3), y_shape=(nc,), root=root, **case['hypers'])


def case_inputs(case):
    rng = np.random.RandomState(9000 + case['seed'])
    n = case['n']
    x0 = rng.random_sample((n, 32, 32, 3))
    y = np.eye(10)[rng.randint(0, 10, n)]
    kc = np.asarray(K_CPTS)[rng.randint(0, len(K_CPTS), n)] if case['dyn'] else None
    return x0, y, kc


def emit(path, draws, seed0):
    sys.path.insert(0, HERE)
    import tf_standin
    import make_ref_graph_golden as M
    sys.modules['tensorflow'] = tf_standin
    sys.path.insert(0, REF)
    import lib.net_types as NT                       # the REFERENCE's modules
    import arch_and_hypers as A
    assert NT.__file__.startswith(REF) and A.__file__.startswith(REF)
    out = {}
    for seed in range(seed0, seed0 + draws):
        case = draw_case(seed)
        tf_standin.reset()
        net = build(A, NT, case)
        rng = np.random.RandomState(seed)
        params = M.ordered_params(net, NT.params_list_rec)
        for name, var in params:
            var.load(M.param_value(name, var.data.shape, rng))
        x0, y, kc = case_inputs(case)
        layers = list(net.layers)
        feed = {net.x0: x0, net.y: y, net.mode: 'tr'}
        if case['tau'] is not None:
            feed[net.τ] = case['tau']
        if kc is not None:
            feed[net.k_cpt] = kc
        if hasattr(layers[0], 'p_tr'):
            snap = [(v, v.data.detach().clone()) for _, v in params]
            vals = tf_standin.run([ℓ.p_tr for ℓ in layers], feed)
            out['%d/p_tr' % seed] = np.stack([np.broadcast_to(np.asarray(v, np.float64), (case['n'],)) for v in vals])
            for v, d in snap:
                v.load(d.numpy())
        feed[net.λ_lrn] = case['lr']
        net.train.run(feed)
        out['%d/after' % seed] = np.array([M.digest(v.data.detach().numpy()) for _, v in params])
        out['%d/names' % seed] = np.array([n for n, _ in params])
    np.savez_compressed(path, **out)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--emit', required=True)
    ap.add_argument('--draws', type=int, default=50)
    ap.add_argument('--seed0', type=int, default=0)
    a = ap.parse_args()
    emit(a.emit, a.draws, a.seed0)
