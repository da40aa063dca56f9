"""GPU parity of whole training / evaluation steps: product nets (HIP kernels through the
C ABI) vs. oracle/ref_net.py (float64 torch-CPU restatement of the reference graph) from
identical injected weights and identical batches.

Tolerances (fp32 kernels vs. a float64 oracle; north_star: routing statistics within 1e-3):
  per-sample costs / probabilities : 2e-4 absolute-or-relative
  gradients, parameter updates      : 1e-4 * max|ref| per tensor, EVERY tensor, no outlier band.  The
                                      oracle is run DECISION-FORCED: the max-pool arg-max and the ReLU
                                      side of every element are read back from the device
                                      (oracle/decisions.py) and the float64 graph differentiates the same
                                      piecewise-linear branch.  (Left to decide for itself, float64
                                      disagrees with fp32 on a near-tie now and then, and one flipped
                                      element moves some gradient tensors by tens of percent.)  The FREE
                                      float64 forward runs too, every step: the decisions the device took may
                                      differ from it on at most max(3, 1e-4 of all decisions) elements
                                      (count_flips; printed) -- forcing cannot hide a systematic decision error.
  BatchNorm moving averages         : 1e-4 relative
Weights are drawn with a fixed seed so every run checks the same numbers.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def batch(n, c0=3, n_cls=10, seed=0, hw=32):
    rng = np.random.default_rng(seed)
    x0 = rng.random((n, hw, hw, c0)).astype(np.float32)
    y = np.eye(n_cls, dtype=np.float32)[rng.integers(0, n_cls, n)]
    return x0, y


def perturb_routers(net, seed=5):
    """The last router map starts at exactly zero (arch_and_hypers.py:49); give it weight so
    routing is non-trivial."""
    rng = np.random.default_rng(seed)
    for ℓ in net.layers:
        if ℓ.router is not None:
            w = ℓ.router.comps[-1].params.w
            w.assign(rng.standard_normal(w.shape) * 0.5)
            b = ℓ.router.comps[-1].params.b
            b.assign(rng.standard_normal(b.shape) * 0.2)


def count_flips(eng, res, n, before):
    """Discrete decisions (2x2 max-pool arg-max on the vert path, ReLU side after BatchNorm) on
    which the fp32 product and the float64 oracle disagree in this forward pass."""
    from oracle import np_ops as O
    flips = total = 0
    for b in eng.blocks:
        pre = res['out'][id(b.conv)]['pre_bn']
        for i in range(b.L):
            s_p = b.s[i][:n].cpu().numpy().astype(np.float64)
            s_o = pre[i].detach().numpy()
            if i < b.L - 1:
                d = O.pool2_argfirst(s_p) != O.pool2_argfirst(s_o)
                flips += int(d.sum()); total += d.size
            if b.has_dz[i]:
                bn = b.bns[i].params
                g, be = (before[id(bn.γ)].cpu().numpy().astype(np.float64), before[id(bn.β)].cpu().numpy().astype(np.float64))
                y_p, _, _ = O.bn_train(s_p, g, be)
                y_o, _, _ = O.bn_train(s_o, g, be)
                d = (y_p > 0) != (y_o > 0)
                flips += int(d.sum()); total += d.size
    return flips, total


TOL = 1e-4          # gradients / updates, relative to the tensor's max |reference|


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (1e-12 + np.abs(b).max())


def run_case(make_net, n, feeds, steps=3, k_cpt_vec=None, c0=3, tol=TOL, n_cls=10, stepper=None, hw=32):
    """Teacher-forced: before every step the oracle is re-synchronised from the product's
    parameters, momentum accumulators and BatchNorm state, so each step checks one
    forward + backward + TALR/momentum update from IDENTICAL state.  (A free-running
    comparison is meaningless: the float64 oracle alone turns a 1e-5 relative weight
    perturbation into a 7-50 % gradient change through max-pool / ReLU flips.)"""
    from oracle.ref_net import RefNet
    net = make_net((hw, hw, c0), (n_cls,))
    eng = net.engine()
    eng.init_params(1234)
    if net._net_kind != 'sr':
        perturb_routers(net)
    ref = RefNet(net)
    lr = 0.05
    # stepper(net) -> callable(feed): how a training step of THIS net is run (default: net.train.run; the co-training
    # tests step it together with other nets, tests/test_cotrain.py)
    train_step = net.train.run if stepper is None else stepper(net)
    worst = {'grad': 0.0, 'update': 0.0}
    flips_total = decisions_total = 0
    for t in range(steps):
        x0, y = batch(n, c0, n_cls, seed=t, hw=hw)
        feed = {net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: lr, **feeds(net, t)}
        kw = {}
        if net._net_kind != 'sr':
            kw['τ'] = feed[net.τ]
        if k_cpt_vec is not None:
            kv = k_cpt_vec(t, n)
            feed[net.k_cpt] = kv
            kw['k_cpt'] = kv
        ref.load_params()
        for p in net._all_params:
            if p.trainable:
                ref.accum[id(p)] = torch.tensor(p.accum.cpu().numpy().astype(np.float64).reshape(p.shape))
        # the engine's G holds the DATA gradient; the L2 term 2*k_l2*mean(p_tr)*w is folded into
        # mpnn_talr_momentum_step, so add it here from the pre-step weights before comparing
        before = {id(p): p.data.clone() for p in net._all_params}
        train_step(feed)
        torch.cuda.synchronize()
        from oracle.decisions import from_product
        # UNFORCED float64 forward first (the oracle still holds the pre-step parameters): the device's own
        # max-pool / ReLU decisions may differ from the free float64 ones only on near-ties.  A systematic
        # wrong-side decision on the device (which the forced run below would faithfully copy) shows up here.
        free = ref.forward(x0, y, 'tr', **{k_: v_ for k_, v_ in kw.items()})
        flips, decisions = count_flips(eng, free, n, before)
        flips_total += flips
        decisions_total += decisions
        assert flips <= max(3, 1e-4 * decisions), ('decision flips vs the free float64 run', flips, decisions, t)
        res = ref.train_step(x0, y, lr, forced=from_product(net, n, before), **kw)
        R = lambda ℓ: res['out'][id(ℓ)]
        for ℓ in net.layers:
            assert np.abs(ℓ.p_tr.cpu().numpy() - R(ℓ)['p_tr'].detach().numpy()).max() < 2e-4, ('p_tr', ℓ.name, t)
            assert np.abs(ℓ.p_ev.cpu().numpy() - R(ℓ)['p_ev'].detach().numpy()).max() == 0, ('p_ev', ℓ.name, t)
        for ℓ in net.leaves:
            ce = R(ℓ)['c_err'].detach().numpy()
            assert np.abs(ℓ.c_err.cpu().numpy() - ce).max() < 2e-4 * (1 + np.abs(ce).max()), ('c_err', t)
            assert np.array_equal(ℓ.δ_cor.cpu().numpy(), R(ℓ)['δ_cor'].numpy()), ('δ_cor', t)
        for ℓ in net.switches:
            rx = R(ℓ.router)['x'].detach().numpy()
            assert np.abs(ℓ.router.x.cpu().numpy() - rx).max() < 2e-4 * (1 + np.abs(rx).max()), ('router.x', t)
        bad, n_checked = [], 0

        def judge(kind, p, err, scale, floor):
            nonlocal n_checked
            n_checked += 1
            worst[kind] = max(worst[kind], float((err - floor) / (scale + 1e-30)))
            if err > tol * scale + floor:
                bad.append((kind, p.owner.name, p.name, float(err), float(scale)))
        for p in net._all_params:
            v0 = before[id(p)].cpu().numpy().astype(np.float64)
            d = p.data.cpu().numpy().astype(np.float64) - v0
            d_ref = ref.V(p).detach().numpy().reshape(-1) - v0
            if not p.trainable:                       # BatchNorm moving averages
                if np.abs(d - d_ref).max() > 1e-4 * np.abs(d_ref).max() + 1e-6:
                    bad.append(('state', p.owner.name, p.name, float(np.abs(d - d_ref).max()), float(np.abs(d_ref).max())))
                continue
            g_ref = res['grads'][id(p)].numpy().reshape(-1)
            g = p.grad.cpu().numpy()
            if p.l2:
                pbar = 1.0 if net._net_kind == 'sr' else float(eng.nodes[p.node].layer.p_tr.mean())
                # (LinTrans(res=True): the penalty pulls towards w_eq, layer_types.py:52)
                g = g + 2 * p.l2 * pbar * (v0 - (np.asarray(p.eq, np.float64).reshape(-1) if p.eq is not None else 0))
            scale = np.abs(g_ref).max()
            # 1e-6 floor: conv biases ahead of BatchNorm have an exactly-zero true gradient
            judge('grad', p, np.abs(g - g_ref).max(), scale, 1e-6)
            judge('update', p, np.abs(d - d_ref).max(), np.abs(d_ref).max(), 1e-7)
        assert not bad, (t, len(bad), 'of', n_checked, bad[:8])
    print('worst relative error over %d steps: gradients %.2e, updates %.2e (tolerance %.0e); '
          'decisions that differ from the free float64 run: %d of %d (%.1e)'
          % (steps, worst['grad'], worst['update'], tol, flips_total, decisions_total, flips_total / max(1, decisions_total)))
    # evaluation pass: moving-average BatchNorm, hard routing, statistics
    x0, y = batch(n, c0, n_cls, seed=99, hw=hw)
    feed = {net.x0: x0, net.y: y, **{k: v for k, v in feeds(net, 0).items()}}
    kw = {}
    if k_cpt_vec is not None:
        feed[net.k_cpt] = k_cpt_vec(0, n)
        kw['k_cpt'] = k_cpt_vec(0, n)
    net.eval(feed)
    st = net.state()
    res = ref.forward(x0, y, 'ev', τ=feed.get(getattr(net, 'τ', None)), **kw)
    rs = ref.stats(res)
    assert np.abs(st[(net, 'acc')].cpu().numpy() - rs['acc']).mean() <= 1e-3
    assert rel(st[(net, 'moc')].cpu().numpy().mean(), rs['moc'].mean()) <= 1e-3
    hist = np.stack([ℓ.p_ev.cpu().numpy() for ℓ in net.leaves]).mean(1)
    assert np.abs(hist - rs['p_leaf'].mean(1)).max() <= 1e-3


def test_sr_chain_3():
    import arch_and_hypers as A
    run_case(A.sr_chain(3), 8, lambda net, t: {})


def test_sr_chain_8():
    import arch_and_hypers as A
    run_case(A.sr_chain(8), 8, lambda net, t: {}, steps=2)


def test_ac_chain():
    import arch_and_hypers as A
    run_case(A.ac_chain(k_cpt=1.6e-8), 16, lambda net, t: {net.τ: A.τ_ds(t * 5000)})


def test_cr_chain():
    import arch_and_hypers as A
    run_case(A.cr_chain(k_cpt=8e-9), 16, lambda net, t: {net.τ: A.τ_cr(t * 5000)})


def test_cr_chain_optimistic_clserr():
    import arch_and_hypers as A
    # (tau = 0.05 pins p_tr of the deep nodes at the epsilon floor: their TALR scale 1/sqrt(mean p_tr^2) is
    # in the hundreds and multiplies gradient errors that sit inside the gradient check's own 1e-6 absolute
    # floor; the worst update error of this case moves between 0.98e-4 and 1.2e-4 with the summation order
    # of the routing kernel, so it gets 2e-4)
    run_case(A.cr_chain(k_cpt=8e-9, optimistic=True, use_cls_err=True), 12, lambda net, t: {net.τ: 0.05}, steps=2, tol=2e-4)


def test_ac_chain_notalr_nokdec():
    import arch_and_hypers as A
    run_case(A.ac_chain(k_cpt=4e-9, talr=False, k_dec=0), 12, lambda net, t: {net.τ: 0.7}, steps=2)


def test_ac_chain_dyn_k_cpt():
    import arch_and_hypers as A
    kv = lambda t, n: np.random.default_rng(t).choice(A.k_cpts, n).astype(np.float32)
    run_case(A.ac_chain(dyn_k_cpt=True), 12, lambda net, t: {net.τ: 0.8}, steps=2, k_cpt_vec=kv)


def test_mnist_sr():
    import arch_and_hypers as A
    run_case(A.sr_chain(2), 8, lambda net, t: {}, steps=2, c0=1)


def test_ragged_batch_37():
    """37 = two 16-row tiles + 5, three 4-image tiles + 1, more than one 32-row group in lin_bwd."""
    import arch_and_hypers as A
    run_case(A.ac_chain(k_cpt=8e-9), 37, lambda net, t: {net.τ: 0.9}, steps=1)


def test_full_batch_128():
    """The benchmark's batch size (arch_and_hypers.py:35)."""
    import arch_and_hypers as A
    run_case(A.cr_chain(k_cpt=1e-9), 128, lambda net, t: {net.τ: 0.1}, steps=1)


def test_bench_config_ac_chain_kcpt0_batch_128():
    """Exactly what bench.py times: ac_chain(k_cpt=0), batch 128, tau and learning rate at t = 0."""
    import arch_and_hypers as A
    run_case(A.ac_chain(k_cpt=0.0), 128, lambda net, t: {net.τ: A.τ_ds(0)}, steps=2)


def test_baseline_cifar10_sr_chain8_batch_128():
    """BASELINE.json `cifar10-sr` at the training batch (scripts/train-nets:81-88: sr_chain(8) is the deepest
    statically-routed net of every *-sr experiment; arch_and_hypers.py:35: batch 128)."""
    import arch_and_hypers as A
    run_case(A.sr_chain(8), 128, lambda net, t: {}, steps=1)


def test_baseline_mnist_sr_chain8_batch_128():
    """BASELINE.json `mnist-sr` as bench.py times it: sr_chain(8) on 32x32x1 inputs (prep-data:35-38 resizes MNIST to
    32x32, one channel), batch 128."""
    import arch_and_hypers as A
    run_case(A.sr_chain(8), 128, lambda net, t: {}, steps=1, c0=1)


def test_baseline_hybrid_dyn_k_cpt_batch_128():
    """BASELINE.json's hybrid adaptive net (scripts/train-adaptive-nets:24-45): ac_chain(dyn_k_cpt=True), a k_cpt per
    sample drawn from k_cpts, batch 128."""
    import arch_and_hypers as A
    kv = lambda t, n: np.random.default_rng(t).choice(A.k_cpts, n).astype(np.float32)
    run_case(A.ac_chain(dyn_k_cpt=True), 128, lambda net, t: {net.τ: A.τ_ds(0)}, steps=1, k_cpt_vec=kv)


def _wide_chain(net_type, widths, n_blocks=4, **hypers):
    """An actor / critic chain of the first `n_blocks` blocks whose routers have hidden layers of `widths` units
    (arch_and_hypers.router with another router_n_chan, or two different widths): LinTrans(n_chan=...) is free in the
    reference (layer_types.py:39-53, arch_and_hypers.py:14,45-49)."""
    import arch_and_hypers as A
    from lib.layer_types import BatchNorm, Chain, LinTrans, Rect, Select

    def router(n_sinks):
        if n_sinks < 2:
            return None
        hidden = []
        for w in widths:
            hidden += [LinTrans(n_chan=w, k_l2=A.k_l2, σ_w=A.σ_w), BatchNorm(), Rect()]
        return Chain(name='Router', comps=[Select(i=-1), *hidden, LinTrans(n_chan=n_sinks, k_l2=A.k_l2, σ_w=0)])

    def rcm(i, *sinks):
        ℓ = A.rcm(i, *sinks)
        ℓ.router = router(len(sinks))
        return ℓ

    def make_net(x0_shape, y_shape):
        node = rcm(n_blocks - 1, A.reg(y_shape[0]))
        for i in range(n_blocks - 2, -1, -1):
            node = rcm(i, A.reg(y_shape[0]), node)
        return net_type(x0_shape=x0_shape, y_shape=y_shape, root=A.pyr(node), **hypers)
    return make_net


def test_wide_router_and_100_classes():
    """Beyond the tuned exit kernels' limits (n_cls <= 16, two equal router layers of <= 16 units): a 32-wide router and
    100 classes, and two DIFFERENT hidden widths (24, 40), run on the any-width forms (csrc/exit_gen.hip) -- same
    tolerances as every other whole-step case, training and evaluation, routed evaluation == dense."""
    from lib.net_types import ActorNet, CriticNet
    import arch_and_hypers as A
    run_case(_wide_chain(ActorNet, (32, 32), k_cpt=1.6e-8), 24, lambda net, t: {net.τ: 0.8}, steps=2, n_cls=100)
    run_case(_wide_chain(CriticNet, (24, 40), k_cpt=8e-9), 20, lambda net, t: {net.τ: 0.3}, steps=2, n_cls=37)
    run_case(_wide_chain(ActorNet, (48, 16), k_cpt=1.6e-8), 160, lambda net, t: {net.τ: 0.8}, steps=1, n_cls=20)     # beyond 128 samples
    net = _wide_chain(ActorNet, (32, 32), k_cpt=1.6e-8)((32, 32, 3), (100,))
    eng = net.engine()
    assert eng.generic_exits
    eng.init_params(5)
    perturb_routers(net)
    rng = np.random.default_rng(8)
    for p in net._all_params:                  # moving averages away from (0, 1), as after training
        if not p.trainable:
            p.assign(rng.random(p.shape) * 0.5 + (0.75 if p.name == 'v_avg' else -0.25))
    for ℓ in net.switches:                     # ... and routers that really decide
        last = ℓ.router.comps[-1].params
        last.w.assign(rng.standard_normal(last.w.shape) * 2.0)
    x0, y = batch(300, 3, 100, seed=2)
    from test_routed_eval import calibrate_exit_fractions
    calibrate_exit_fractions(net, x0, y, [1 / 4] * 3)          # a quarter of the batch leaves at each of the four exits
    net.eval({net.x0: x0, net.y: y})
    dense = [ℓ.p_ev.clone() for ℓ in net.layers]
    ce = [ℓ.c_err.clone() for ℓ in net.leaves]
    assert [float(ℓ.p_ev.mean()) for ℓ in net.leaves] == [0.25] * 4
    net.eval({net.x0: x0, net.y: y}, routed=1)
    assert all(torch.equal(a, ℓ.p_ev) for a, ℓ in zip(dense, net.layers))
    for c, ℓ in zip(ce, net.leaves):
        reach = ℓ.p_ev > 0
        assert torch.equal(c[reach], ℓ.c_err[reach])


def test_tuned_and_any_width_exit_kernels_agree():
    """The shipped chain on the any-width forms (MPNN_GENERIC_EXITS=1) against the tuned kernels: the same training
    steps to fp32 summation order."""
    import os
    import arch_and_hypers as A
    outs = []
    for gen in ('0', '1'):
        os.environ['MPNN_GENERIC_EXITS'] = gen
        try:
            net = A.ac_chain(k_cpt=1.6e-8, seed=3)((32, 32, 3), (10,))
            eng = net.engine()
        finally:
            del os.environ['MPNN_GENERIC_EXITS']
        assert eng.generic_exits == (gen == '1')
        perturb_routers(net)
        x0, y = batch(48, seed=7)
        for _ in range(2):
            net.train.run({net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: 0.05, net.τ: 0.8})
        torch.cuda.synchronize()
        outs.append((eng.P.cpu().numpy().copy(), [ℓ.p_tr.cpu().numpy().copy() for ℓ in net.layers]))
    (Pa, pa), (Pb, pb) = outs
    assert np.abs(Pa - Pb).max() <= 3e-4 * np.abs(Pa).max(), np.abs(Pa - Pb).max()
    for u, v in zip(pa, pb):
        assert np.abs(u - v).max() <= 2e-4


def test_batch_of_one_eval():
    """A single image through the evaluation path (BatchNorm moving averages; batch statistics of
    one sample would be degenerate in 'tr')."""
    import arch_and_hypers as A
    from oracle.ref_net import RefNet
    net = A.ac_chain(k_cpt=0.0)((32, 32, 3), (10,))
    eng = net.engine()
    eng.init_params(11)
    perturb_routers(net)
    ref = RefNet(net)
    ref.load_params()
    x0, y = batch(1, seed=4)
    net.eval({net.x0: x0, net.y: y})
    res = ref.forward(x0, y, 'ev')
    for ℓ in net.leaves:
        ce = res['out'][id(ℓ)]['c_err'].detach().numpy()
        assert np.abs(ℓ.c_err.cpu().numpy() - ce).max() < 2e-4 * (1 + np.abs(ce).max())
        assert np.array_equal(ℓ.p_ev.cpu().numpy(), res['out'][id(ℓ)]['p_ev'].numpy())


def test_known_answers_at_init():
    """KA3/KA4: at initialisation every router output is exactly 0, so the test-time
    routing histogram is [1,0,...,0], moc = 1 368 608, leaf p_tr ~ 2^-(j+1)."""
    import arch_and_hypers as A
    net = A.ac_chain(k_cpt=0.0)((32, 32, 3), (10,))
    x0, y = batch(16)
    net.eval({net.x0: x0, net.y: y})
    st = net.state()
    leaves = list(net.leaves)
    hist = [float(ℓ.p_ev.mean()) for ℓ in leaves]
    assert hist == [1.0] + [0.0] * 7
    assert float(st[(net, 'moc')].mean()) == 1361664 + 4384 + 2560
    ptr = [float(ℓ.p_tr.mean()) for ℓ in leaves]
    assert abs(sum(ptr) - 1) < 1e-6
    for j, p in enumerate(ptr):
        assert abs(p - 2.0 ** -(min(j, 6) + 1)) < 1e-5


def test_training_batch_256():
    """A training batch beyond the 128 samples of arch_and_hypers.py:35 (the reference's placeholders are
    (None, ...), net_types.py:50-51): the router tails run their any-size forms (exit_tail.hip, router_*_big)."""
    import arch_and_hypers as A
    run_case(A.ac_chain(k_cpt=1.6e-8), 256, lambda net, t: {net.τ: 0.8}, steps=1)


def test_training_batch_512_strip_bodies():
    """512 samples per step: the forward convs of the big maps take the wave-per-strip bodies (conv_strip.h, capacity
    >= 512) in TRAINING mode too -- batch-statistics BatchNorm on load, statistics of the outputs from the strips."""
    import arch_and_hypers as A
    run_case(A.ac_chain(k_cpt=1.6e-8), 512, lambda net, t: {net.τ: 0.8}, steps=1)


def test_forward_only_fetch_in_tr_mode():
    """A fetch with mode 'tr' and no train op (net_types.py:50-52; the placeholders accept it): batch-statistics
    BatchNorm, soft routing p_tr -- and, as in the reference (layer_types.py:231-236), the consumed BatchNorms move
    their averages while parameters, momentum and gradients stay untouched."""
    import arch_and_hypers as A
    from oracle.ref_net import RefNet
    net = A.ac_chain(k_cpt=1.6e-8)((32, 32, 3), (10,))
    eng = net.engine()
    eng.init_params(1234)
    perturb_routers(net)
    ref = RefNet(net)
    ref.load_params()
    n = 16
    x0, y = batch(n, seed=3)
    P0, A0, S0 = eng.P.clone(), eng.A.clone(), eng.S.clone()
    for rep in range(2):                                   # (twice: the second run starts from cleared accumulators)
        ref.load_params()
        net.eval({net.x0: x0, net.y: y, net.mode: 'tr', net.τ: 0.9})
        torch.cuda.synchronize()
        res = ref.forward(x0, y, 'tr', τ=0.9)
        R = lambda ℓ: res['out'][id(ℓ)]
        for ℓ in net.layers:
            assert np.abs(ℓ.p_tr.cpu().numpy() - R(ℓ)['p_tr'].detach().numpy()).max() < 2e-4
        for ℓ in net.leaves:
            ce = R(ℓ)['c_err'].detach().numpy()
            assert np.abs(ℓ.c_err.cpu().numpy() - ce).max() < 2e-4 * (1 + np.abs(ce).max())
        assert torch.equal(eng.P, P0) and torch.equal(eng.A, A0)
        moved = 0
        for b in eng.blocks:
            for i, bn in enumerate(b.bns):
                m, v = bn.params.m_avg.numpy(), bn.params.v_avg.numpy()
                m_ref, v_ref = ref.state[id(bn.params.m_avg)].numpy(), ref.state[id(bn.params.v_avg)].numpy()
                if b.has_dz[i]:
                    m_ref, v_ref = (t.numpy() for t in R(bn)['new_avg'])
                    moved += 1
                assert np.abs(m - m_ref).max() <= 1e-4 * (1e-3 + np.abs(m_ref).max()), (b.H[i], i)
                assert np.abs(v - v_ref).max() <= 1e-4 * (1e-3 + np.abs(v_ref).max()), (b.H[i], i)
            if b.router is not None:
                for k in (2, 5):
                    bn = b.router.comps[k]
                    m_ref, v_ref = (t.numpy() for t in res['out'][id(bn)]['new_avg'])
                    assert np.abs(bn.params.m_avg.numpy() - m_ref).max() <= 1e-4 * (1e-3 + np.abs(m_ref).max())
                    assert np.abs(bn.params.v_avg.numpy() - v_ref).max() <= 1e-4 * (1e-3 + np.abs(v_ref).max())
        assert moved >= 10 and not torch.equal(eng.S, S0)
    # ... and training goes on from there
    net.train.run({net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: 0.05, net.τ: 0.9})
    assert torch.isfinite(eng.P).all() and not torch.equal(eng.P, P0)
