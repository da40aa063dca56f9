"""GPU: the any-width exit path (csrc/exit_gen.hip, rewritten on MFMA tiles in round 5) on randomly drawn head and router
widths -- class counts 2 ... 1 000, hidden layers of 1 ... 256 units (odd, prime, equal and unequal pairs), 2 ... 4 blocks,
batches of 5 ... 200 samples (on both sides of the tuned tails' 128) -- whole training steps against the float64 oracle,
decision-forced, at the whole-net tolerances of tests/test_net_parity.py::run_case; then evaluation, dense == routed."""
import os

import numpy as np
import pytest
import torch

from test_net_parity import _wide_chain, batch, perturb_routers, run_case

pytestmark = pytest.mark.gpu

SEEDS = [int(s) for s in os.environ.get('MPNN_FUZZ_WIDTH_SEEDS', '0 1 2 3 4 5').split()]
WIDTHS = [1, 2, 3, 7, 8, 13, 16, 17, 24, 31, 32, 33, 48, 64, 100, 127, 128, 200, 256]
CLASSES = [2, 3, 5, 10, 11, 16, 17, 37, 100, 257, 1000]
BATCHES = [5, 9, 16, 33, 64, 127, 128, 129, 200]          # (two-sample BatchNorm statistics amplify fp32 rounding beyond the 1e-4 gate)


@pytest.mark.parametrize('seed', SEEDS)
def test_any_width_exits_on_random_widths(seed):
    from lib.net_types import ActorNet, CriticNet
    rng = np.random.default_rng(700 + seed)
    kind = (ActorNet, CriticNet)[int(rng.integers(0, 2))]
    w1 = int(rng.choice(WIDTHS))
    w2 = w1 if rng.random() < 0.3 else int(rng.choice(WIDTHS))
    n_cls, n = int(rng.choice(CLASSES)), int(rng.choice(BATCHES))
    if n_cls >= 257 and n > 64:
        n = 33                                            # (the float64 oracle's time)
    blocks = int(rng.integers(2, 5))
    hyp = dict(k_cpt=float(rng.choice([0.0, 4e-9, 1.6e-8])))
    if kind is CriticNet and rng.random() < 0.5:
        hyp['optimistic'] = True
    print('seed %d: %s, router %d-%d, %d classes, %d blocks, batch %d, %s' % (seed, kind.__name__, w1, w2, n_cls, blocks, n, hyp))
    mk = _wide_chain(kind, (w1, w2), n_blocks=blocks, **hyp)
    run_case(mk, n, lambda net, t: {net.τ: 0.6}, steps=1, n_cls=n_cls)
    # evaluation: the routed pass (every prefix) against the dense one
    net = mk((32, 32, 3), (n_cls,))
    eng = net.engine()
    eng.init_params(5)
    perturb_routers(net)
    x0, y = batch(max(n, 40), 3, n_cls, seed=seed)
    net.eval({net.x0: x0, net.y: y})
    dense = [ℓ.p_ev.clone() for ℓ in net.layers]
    ce = [ℓ.c_err.clone() for ℓ in net.leaves]
    for routed in (1, 2, True):
        net.eval({net.x0: x0, net.y: y}, routed=routed)
        assert all(torch.equal(a, ℓ.p_ev) for a, ℓ in zip(dense, net.layers)), routed
        for c, ℓ in zip(ce, net.leaves):
            reach = ℓ.p_ev > 0
            assert torch.equal(c[reach], ℓ.c_err[reach]), routed
