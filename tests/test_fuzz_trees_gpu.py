"""GPU: randomly drawn TREES -- forks of two or three sinks, STATIC links (a block whose only sink is another block: no
router, no exit; its child inherits the reach of the nearest switch above), leaves at any depth -- beyond the fixed
tree shapes of the other tests.  Whole training steps against the float64 oracle (decision-forced,
tests/test_net_parity.py::run_case), then evaluation: routed == dense for every prefix depth (the prefix walk's
reach logic through static links: lib/_plan.py:_program_ev, csrc/exit_ev.hip), sample lists == nonzero(p_ev)."""
import os

import numpy as np
import pytest
import torch

from test_net_parity import batch, perturb_routers, run_case

pytestmark = pytest.mark.gpu

SEEDS = [int(s) for s in os.environ.get('MPNN_FUZZ_TREE_SEEDS', '0 1 2 3').split()]


def draw_tree(rng, max_depth):
    """Nested tuples (has_leaf, [children]) for the block at each position."""
    def node(i):
        k = int(rng.choice([0, 1, 1, 2])) if i < max_depth else 0
        has_leaf = True if k == 0 else bool(rng.random() < (0.55 if k == 1 else 0.75))
        return (has_leaf, [node(i + 1) for _ in range(k)])
    return node(0)


def describe(t):
    return ('L' if t[0] else '') + ('(' + ','.join(describe(c) for c in t[1]) + ')' if t[1] else '')


def make_tree(net_type, spec, **hypers):
    import arch_and_hypers as A

    def build(t, i, n_cls):
        sinks = ([A.reg(n_cls)] if t[0] else []) + [build(c, i + 1, n_cls) for c in t[1]]
        return A.rcm(i, *sinks)

    def make_net(x0_shape, y_shape):
        return net_type(x0_shape=x0_shape, y_shape=y_shape, root=A.pyr(build(spec, 0, y_shape[0])), **hypers)
    return make_net


@pytest.mark.parametrize('seed', SEEDS)
def test_random_trees(seed):
    from lib.net_types import ActorNet, CriticNet
    from test_routed_eval import check_routed_equals_dense, randomise_routers
    rng = np.random.default_rng(1300 + seed)
    has_switch = lambda t: (int(t[0]) + len(t[1]) >= 2) or any(has_switch(c) for c in t[1])
    spec = draw_tree(rng, int(rng.integers(2, 5)))
    while not has_switch(spec):                       # (a chain without a switch is statically routed: covered elsewhere)
        spec = draw_tree(rng, int(rng.integers(2, 5)))
    kind = (ActorNet, CriticNet)[int(rng.integers(0, 2))]
    n = int(rng.choice([6, 24, 128]))
    print('seed %d: %s, tree %s, batch %d' % (seed, kind.__name__, describe(spec), n))
    mk = make_tree(kind, spec, k_cpt=float(rng.choice([0.0, 4e-9])))
    run_case(mk, n, lambda net, t: {net.τ: 0.7}, steps=2)
    net = mk((32, 32, 3), (10,))
    eng = net.engine()
    eng.init_params(11)
    g = np.random.default_rng(seed)
    for p in net._all_params:
        if not p.trainable:
            p.assign(g.random(p.shape) * 0.5 + (0.75 if p.name == 'v_avg' else -0.25))
    randomise_routers(net, seed=seed, scale=1.0)
    x0, y = batch(150, seed=seed + 1)
    depth = max(eng._depths().values()) + 1
    check_routed_equals_dense(net, x0, y, modes=tuple([True] + list(range(1, depth + 2))))
