"""Tree-structured nets (scripts/arch_and_hypers.py:99-127): blocks with several child
blocks, 3-way switches.  What a tree adds to the chain path:

  * a parent map's gradient is the SUM over its child blocks' input-gradient convs
    (mpnn_dgrad_horz_args.accumulate), with the ReLU mask / BatchNorm reductions linear in it;
  * mpnn_route / mpnn_exit_tail with 3 sinks per switch; 95 routing-tree nodes (32 samples per
    routing workgroup);
  * routed evaluation with one sample list per child.

A small tree (3-way switch over two sub-chains) is checked step by step against the decision-forced
float64 oracle exactly like the chains (tests/test_net_parity.py); the full 47-block ac_tree / cr_tree
of the reference run a training step against the oracle at a tiny batch, and dense == routed
evaluation.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_net_parity import run_case, batch, perturb_routers


def small_tree(type_, **hypers):
    import arch_and_hypers as A

    def make_net(x0_shape, y_shape):
        nc = y_shape[0]
        root = A.pyr(A.rcm(0, A.reg(nc),
                           A.rcm(1, A.reg(nc), A.rcm(2, A.reg(nc))),
                           A.rcm(1, A.reg(nc), A.rcm(2, A.reg(nc), A.rcm(3, A.reg(nc))))))
        return type_(x0_shape=x0_shape, y_shape=y_shape, root=root, **hypers)
    return make_net


def test_small_actor_tree_vs_oracle():
    from lib.net_types import ActorNet
    run_case(small_tree(ActorNet, k_cpt=1.6e-8), 12, lambda net, t: {net.τ: 0.8}, steps=2)


def test_small_critic_tree_vs_oracle():
    from lib.net_types import CriticNet
    run_case(small_tree(CriticNet, k_cpt=8e-9, optimistic=True), 12, lambda net, t: {net.τ: 0.05}, steps=2)


@pytest.mark.parametrize('kind', ['ac', 'cr'])
def test_reference_tree_one_step_vs_oracle(kind):
    """ac_tree / cr_tree of arch_and_hypers.py:99-139: 47 blocks, 47 leaves, 39 switches."""
    import arch_and_hypers as A
    mk = A.ac_tree(k_cpt=4e-9) if kind == 'ac' else A.cr_tree(k_cpt=4e-9)
    # 5e-4 instead of 1e-4: eight levels of routing products put p_tr at 1e-4..1e-2, and the actor's
    # dL/dr = softmax_j (u_j - sum_i softmax_i u_i) / tau cancels to a few digits when the children's
    # values are close (measured worst 1.3e-4 on four router tensors of 1834 checks; the rest < 1e-4)
    run_case(mk, 6, lambda net, t: {net.τ: 0.5 if kind == 'ac' else 0.05}, steps=1, tol=5e-4)


def test_reference_tree_one_step_at_the_training_batch():
    """ac_tree (arch_and_hypers.py:99-127) at the training batch of arch_and_hypers.py:35: one full step of the
    47-block tree at n = 128 against the oracle (the small cases above run at n = 6 and 12)."""
    import arch_and_hypers as A
    run_case(A.ac_tree(k_cpt=4e-9), 128, lambda net, t: {net.τ: 0.5}, steps=1, tol=5e-4)


def test_reference_tree_routed_eval_equals_dense():
    import arch_and_hypers as A
    from test_routed_eval import check_routed_equals_dense
    net = A.ac_tree(k_cpt=1e-9)((32, 32, 3), (10,))
    eng = net.engine()
    eng.init_params(5)
    assert len(eng.blocks) == 47 and len(eng.leaves) == 47 and len(eng.switches) == 39 and eng.max_sinks == 3
    x0, y = batch(200, seed=2)
    net.eval({net.x0: x0, net.y: y})                       # KA3 on the tree: zero routers -> everything leaves at exit 0
    hist = [float(nd.layer.p_ev.mean()) for nd in eng.leaves]
    assert hist[0] == 1.0 and sum(hist) == 1.0
    perturb_routers(net, seed=8)
    dense = check_routed_equals_dense(net, x0, y)
    hist = np.stack([dense['p_ev'][nd.idx] for nd in eng.leaves]).mean(1)
    assert abs(hist.sum() - 1) < 1e-6 and (hist > 0).sum() >= 6, hist
