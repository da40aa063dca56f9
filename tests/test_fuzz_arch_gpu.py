"""GPU: randomly drawn ARCHITECTURES -- the shipped specs all use arch_and_hypers.arch (4 x 16, 4 x 16, 3 x 32, 3 x 32,
2 x 64, 2 x 64, 128, 128 channels per scale and block); the reference's MultiscaleConvMax takes any n_chan list
(scripts/lib/layer_types.py:277-297).  Chains of 2 - 6 blocks with 16 / 32 / 64 / 128 channels per scale, 1 - 4 scales
(non-increasing along the chain), on 3- and 1-channel images, statically and dynamically routed: whole training steps
against the float64 oracle, decision-forced (tests/test_net_parity.py::run_case), then dense == routed evaluation.  An
architecture the engine refuses must be refused LOUDLY (NotImplementedError), never miscomputed."""
import os

import numpy as np
import pytest
import torch

from test_net_parity import batch, perturb_routers, run_case

pytestmark = pytest.mark.gpu

SEEDS = [int(s) for s in os.environ.get('MPNN_FUZZ_ARCH_SEEDS', '0 1 2 3').split()]


def draw_arch(rng):
    blocks = int(rng.integers(2, 7))
    L = int(rng.integers(2, 5))
    ch = int(rng.choice([16, 16, 32]))
    arch = []
    for b in range(blocks):
        if b and rng.random() < 0.45 and L > 1:
            L -= 1
        if b and rng.random() < 0.5 and ch < 128:
            ch *= 2
        arch.append(L * [ch])
    return arch


def make_chain(net_type, arch, n_scales, **hypers):
    import arch_and_hypers as A
    from lib.layer_types import (Chain, MultiscaleBatchNorm, MultiscaleConvMax, MultiscaleRect, ToPyramid)

    def rcm(i, *sinks):
        body = [MultiscaleConvMax(n_chan=arch[i], supp=3, k_l2=A.k_l2, σ_w=A.σ_w), MultiscaleBatchNorm(), MultiscaleRect()]
        return Chain(name='ReConvMax', sinks=sinks, router=A.router(len(sinks)), comps=body)

    def make_net(x0_shape, y_shape):
        sr = net_type.__name__ == 'SRNet'
        node = rcm(len(arch) - 1, A.reg(y_shape[0]))
        for i in range(len(arch) - 2, -1, -1):
            node = rcm(i, node) if sr else rcm(i, A.reg(y_shape[0]), node)
        root = Chain(name='ToPyramid', sinks=[node], router=None, comps=[ToPyramid(n_scales=n_scales)])
        return net_type(x0_shape=x0_shape, y_shape=y_shape, root=root, **hypers)
    return make_net


@pytest.mark.parametrize('seed', SEEDS)
def test_random_architectures(seed):
    from lib.net_types import ActorNet, CriticNet, SRNet
    rng = np.random.default_rng(900 + seed)
    arch = draw_arch(rng)
    kind = (ActorNet, CriticNet, SRNet)[int(rng.integers(0, 3))]
    c0 = int(rng.choice([3, 3, 1]))
    n = int(rng.choice([6, 16, 40, 128]))
    hw = int(os.environ.get('MPNN_FUZZ_ARCH_HW', '32'))
    hyp = {} if kind is SRNet else dict(k_cpt=float(rng.choice([0.0, 4e-9])))
    print('seed %d: %s, arch %s, %d input channels, batch %d' % (seed, kind.__name__, arch, c0, n))
    mk = make_chain(kind, arch, len(arch[0]), **hyp)
    feeds = (lambda net, t: {}) if kind is SRNet else (lambda net, t: {net.τ: 0.7})
    try:
        run_case(mk, n, feeds, steps=2, c0=c0, hw=hw)
    except NotImplementedError as e:
        pytest.skip('refused loudly: %s' % e)
    if kind is SRNet:
        return
    net = mk((hw, hw, c0), (10,))
    net.engine().init_params(5)
    perturb_routers(net)
    x0, y = batch(200, c0, 10, seed=seed, hw=hw)
    net.eval({net.x0: x0, net.y: y})
    dense = [ℓ.p_ev.clone() for ℓ in net.layers]
    ce = [ℓ.c_err.clone() for ℓ in net.leaves]
    for routed in (1, 2, True):
        net.eval({net.x0: x0, net.y: y}, routed=routed)
        assert all(torch.equal(a, ℓ.p_ev) for a, ℓ in zip(dense, net.layers)), routed
        for c, ℓ in zip(ce, net.leaves):
            reach = ℓ.p_ev > 0
            assert torch.equal(c[reach], ℓ.c_err[reach]), routed
