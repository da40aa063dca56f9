"""GPU: the multi-stream DAG schedule and hipGraph replay give the same step as the
sequential eager order (same kernels, same inputs; only atomics ordering may differ)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def run(multi, graph, steps=4):
    import arch_and_hypers as A
    net = A.ac_chain(k_cpt=1.6e-8, seed=7)((32, 32, 3), (10,))
    eng = net.engine()
    eng.multi_stream, eng.use_graph = multi, graph
    rng = np.random.default_rng(3)
    x0 = rng.random((32, 32, 32, 3)).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, 32)]
    for t in range(steps):
        net.train.run({net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: 0.05, net.τ: 1.0})
    torch.cuda.synchronize()
    return eng.P.cpu().numpy().copy(), eng.S.cpu().numpy().copy()


def test_multi_stream_and_graph_match_sequential():
    """ONE step from the same state: every schedule must agree to rounding (the multi-stream schedule uses
    the unfused backward launches, i.e. other summation trees: 1e-7).  A wrong dependency shows up here.
    Over several steps only runs of the SAME launch list are compared (graph replay vs eager): training
    amplifies a 1e-7 difference through ReLU / max-pool decisions -- 2e-4 after the second step, 6e-4 after
    the fourth, measured -- so a multi-step comparison ACROSS schedules tests the net's conditioning, not
    the scheduler."""
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    p0, s0 = run(False, False, steps=1)
    for multi, graph in ((True, False), (False, True), (True, True)):
        p1, s1 = run(multi, graph, steps=1)
        assert rel(p1, p0) <= 2e-6 and rel(s1, s0) <= 2e-6, (multi, graph, rel(p1, p0), rel(s1, s0))
    p0, s0 = run(False, False)
    p1, s1 = run(False, True)
    assert rel(p1, p0) <= 1e-6 and rel(s1, s0) <= 1e-6, ('graph replay vs eager, 4 steps', rel(p1, p0))
    p0, s0 = run(True, False)
    p1, s1 = run(True, True)
    assert rel(p1, p0) <= 1e-4 and rel(s1, s0) <= 1e-4, ('multi-stream: graph replay vs eager, 4 steps', rel(p1, p0))


def _net128(seed=5):
    import arch_and_hypers as A
    net = A.ac_chain(k_cpt=1.6e-8, seed=seed)((32, 32, 3), (10,))
    rng = np.random.default_rng(0)
    x0 = rng.random((128, 32, 32, 3)).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, 128)]
    return net, {net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: 0.05, net.τ: 1.0}


def test_training_step_is_repeatable():
    """The same step from the same state gives the same gradients and parameters, run to run
    (only the fp32 atomics of the exit-path dW may reorder: ~1e-8).  This caught a stale
    MFMA-accumulator read on gfx950 (common.h: mfma_drain) that made whole tile rows of a conv
    differ by O(1) between launches."""
    net, feed = _net128()
    eng = net.engine()
    for _ in range(2):
        net.train.run(feed)
    torch.cuda.synchronize()
    P0, A0, S0 = eng.P.clone(), eng.A.clone(), eng.S.clone()
    outs = []
    for rep in range(4):
        eng.P.copy_(P0); eng.A.copy_(A0); eng.S.copy_(S0); eng.invalidate_packs()
        net.train.run(feed)
        torch.cuda.synchronize()
        outs.append((eng.G.clone(), eng.P.clone(), [s.clone() for b in eng.blocks for s in b.s]))
    g0, p0, s0 = outs[0]
    for g, p, s in outs[1:]:
        for a, b in zip(s, s0):
            assert torch.equal(a, b), 'forward conv sums differ between identical launches'
        assert (g - g0).abs().max().item() <= 1e-6 * g0.abs().max().item()
        assert (p - p0).abs().max().item() <= 1e-6 * p0.abs().max().item()


def test_wavefront_groups_equal_single_launches():
    """mpnn_msconv_fwd_group (one launch per wavefront level) computes what one mpnn_msconv_fwd
    launch per conv computes.  Block 0 (no BatchNorm on its input) is bit-identical; deeper blocks
    see the fp64 statistics slots filled in a different workgroup order, i.e. fp32 coefficients
    that may differ in the last bit."""
    net, feed = _net128(seed=11)
    eng = net.engine()
    eng.use_graph = False
    res = {}
    for group in (True, False):
        eng.group_fwd = group
        eng.init_params(11)
        net.train.run(feed)
        torch.cuda.synchronize()
        res[group] = [s.clone() for b in eng.blocks for s in b.s] + [s.clone() for b in eng.blocks for s in b.sp]
    kinds = {op.what for op in eng.program('tr', 128)['fwd']}
    assert 'msconv_fwd' in kinds and 'fwd_group' not in kinds
    n0 = eng.blocks[0].L
    for k, (a, b) in enumerate(zip(res[True], res[False])):
        if k < n0:
            assert torch.equal(a, b), k
        assert (a - b).abs().max().item() <= 1e-4 * b.abs().max().item(), k


def test_training_memorises_fixed_batches_and_stays_finite():
    """1500 graph-replayed steps over 4 fixed random batches with the driver's schedules: parameters
    stay finite and the routed net classifies the batches it has seen (a functional check of the whole
    step -- forward, exits, router, backward, TALR/momentum -- beyond per-step parity)."""
    import arch_and_hypers as A
    net = A.ac_chain(k_cpt=1.6e-8, seed=3)((32, 32, 3), (10,))
    eng = net.engine()
    g = torch.Generator(device='cuda').manual_seed(0)
    xs = torch.rand((4, 128, 32, 32, 3), device='cuda', generator=g)
    ys = torch.eye(10, device='cuda')[torch.randint(0, 10, (4, 128), device='cuda', generator=g)]
    for t in range(1500):
        net.train.run({net.x0: xs[t % 4], net.y: ys[t % 4], net.mode: 'tr', net.λ_lrn: A.λ_lrn(t), net.τ: A.τ_ds(t)})
    torch.cuda.synchronize()
    assert torch.isfinite(eng.P).all() and torch.isfinite(eng.S).all()
    net.eval({net.x0: xs[0], net.y: ys[0]})
    acc = float(net.state()[(net, 'acc')].mean())
    assert acc > 0.9, acc


def test_optimizer_keeps_the_weight_packs_current():
    """mpnn_talr_momentum_step writes every updated conv weight into its slots of the forward and backward
    packs: after training steps (eager and graph replay) the packs must equal what mpnn_pack_weights
    builds from the parameters, bit for bit; a Param.assign makes the engine rebuild them."""
    import arch_and_hypers as A
    net = A.ac_chain(k_cpt=1e-8, seed=3)((32, 32, 3), (10,))
    eng = net.engine()
    rng = np.random.default_rng(0)
    x0 = rng.random((32, 32, 32, 3)).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, 32)]
    for t in range(5):
        net.train.run({net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: 0.05, net.τ: 1.0})
    torch.cuda.synchronize()
    assert eng._packs_fresh
    kept = eng.packs.clone()
    eng._pack()
    torch.cuda.synchronize()
    assert torch.equal(kept, eng.packs)
    w = eng.blocks[3].conv.params.w_horz_0
    w.assign(w.numpy() * 0.5)
    assert not eng._packs_fresh
    net.eval({net.x0: x0, net.y: y})
    torch.cuda.synchronize()
    kept = eng.packs.clone()
    eng._pack()
    torch.cuda.synchronize()
    assert eng._packs_fresh and torch.equal(kept, eng.packs)


def test_backward_dependency_levels():
    """The backward schedule (Engine._bwd_schedule): the 20 (block, scale) triples of the 8-block chain are 16
    launches, the four two-member levels pair a large map of block b+1 with a small map of block b; no launch holds
    two members that write the same map (tree nets: siblings accumulate into one parent map); and the level
    launches give bit-identical results to one launch per triple."""
    import numpy as np
    import torch
    import arch_and_hypers as A
    net = A.ac_chain(k_cpt=1.6e-8)((32, 32, 3), (10,))
    eng = net.engine()
    eng.init_params(7)
    order = [(kb, b, i) for kb, b in enumerate(reversed(eng.blocks)) for i in range(b.L - 1, -1, -1)]
    groups = eng._bwd_schedule(order, 128)
    assert sum(len(g) for g in groups) == 20 and len(groups) == 16
    pairs = [sorted((m[1].H[m[2]] for m, _ in g), reverse=True) for g in groups if len(g) > 1]
    assert pairs == [[16, 4], [16, 4], [16, 4], [32, 8]]
    seen = set()
    for g in groups:                                   # a topological order of Engine._bwd_deps
        for (kb, b, i), _ in g:
            assert all((id(d), j) in seen for d, j in eng._bwd_deps(b, i))
        seen |= {(id(b), i) for (kb, b, i), _ in g}
    tree = A.ac_tree(k_cpt=1e-9)((32, 32, 3), (10,))
    te = tree.engine()
    order = [(kb, b, i) for kb, b in enumerate(reversed(te.blocks)) for i in range(b.L - 1, -1, -1)]
    for g in te._bwd_schedule(order, 16):
        tg = [(id(b.parent), b.in_map[i]) for (kb, b, i), _ in g if b.parent is not None]
        assert len(tg) == len(set(tg)) and len(g) <= 4
    # levels on / off: the same arithmetic in the same order inside every body -> identical parameters
    rng = np.random.default_rng(0)
    x0 = rng.random((64, 32, 32, 3)).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, 64)]
    out = []
    for levels in (True, False):
        n2 = A.ac_chain(k_cpt=1.6e-8)((32, 32, 3), (10,))
        e2 = n2.engine()
        e2.init_params(7)
        e2.bwd_levels = levels
        n2.train.run({n2.x0: x0, n2.y: y, n2.mode: 'tr', n2.λ_lrn: 0.05, n2.τ: 1.0})
        torch.cuda.synchronize()
        out.append(e2.P.clone())
    # ONE step (later steps amplify rounding chaotically); the weight-gradient split differs between the two
    # schedules, so the slab sums round differently: agreement to fp32 rounding of the update, not bit for bit
    assert float((out[0] - out[1]).abs().max()) <= 2e-6 * float(out[1].abs().max())


@pytest.mark.parametrize('kind', ['ac', 'cr', 'sr', 'dyn', 'dyn_dev', 'dyn_cr_wide'])
def test_k_steps_in_one_graph_equal_k_single_steps(kind):
    """Engine.run_steps: K training steps as ONE hipGraph replay -- the schedule values of step j (learning rate,
    temperature: different in every step here) reach the step through the device ring and the head workgroup of its
    mpnn_exit_tail_fwd -- against K single-step replays from the same state: the same launches on the same data, so the
    parameters agree to the last bits of the fp64-atomic statistics (1e-6), call after call (warm-up, capture, replays).
    dyn*: per-sample k_cpt vectors (dyn_k_cpt nets, train-adaptive-nets), a different one in every step -- from the host
    (dyn), as device tensors (dyn_dev), and on the any-width exit kernels (dyn_cr_wide: 20 classes)."""
    import arch_and_hypers as A
    mk = {'ac': lambda: A.ac_chain(k_cpt=1.6e-8, seed=7), 'cr': lambda: A.cr_chain(k_cpt=8e-9, seed=7), 'sr': lambda: A.sr_chain(8),
          'dyn': lambda: A.ac_chain(dyn_k_cpt=True, seed=7), 'dyn_dev': lambda: A.ac_chain(dyn_k_cpt=True, seed=7),
          'dyn_cr_wide': lambda: A.cr_chain(dyn_k_cpt=True, seed=7)}[kind]
    n_cls = 20 if kind == 'dyn_cr_wide' else 10
    nets = [mk()((32, 32, 3), (n_cls,)) for _ in range(2)]
    for net in nets:
        net.engine().init_params(77)
    n, K = 32, 4
    rng = np.random.default_rng(3)
    x0 = torch.from_numpy(rng.random((n, 32, 32, 3)).astype(np.float32)).cuda()
    y = torch.from_numpy(np.eye(n_cls, dtype=np.float32)[rng.integers(0, n_cls, n)]).cuda()
    engs = [net.engine() for net in nets]
    for e in engs:
        e._ensure_capacity(n)
        e.x0[:n].copy_(x0); e.y[:n].copy_(y)

    def feed(net, t):
        e = net.engine()
        f = {net.x0: e.x0[:n], net.y: e.y[:n], net.mode: 'tr', net.λ_lrn: 0.05 / (1 + 0.3 * t)}
        if kind != 'sr':
            f[net.τ] = 1.0 / (1 + 0.1 * t)
        if kind.startswith('dyn'):
            kv = np.random.default_rng(100 + t).choice(A.k_cpts, n).astype(np.float32)
            f[net.k_cpt] = torch.from_numpy(kv).cuda() if kind == 'dyn_dev' else kv
        return f
    a, b = nets
    rel = lambda u, v: float((u - v).abs().max() / v.abs().max())
    for call in range(4):
        ts = range(call * K, (call + 1) * K)
        a.train.run_steps([feed(a, t) for t in ts])
        for t in ts:
            b.train.run(feed(b, t))
        torch.cuda.synchronize()
        assert rel(engs[0].P, engs[1].P) <= 1e-6 and rel(engs[0].A, engs[1].A) <= 1e-6 and rel(engs[0].S, engs[1].S) <= 1e-6, (call,)
        for la, lb in zip(a.layers, b.layers):
            assert torch.equal(la.p_ev, lb.p_ev) and torch.allclose(la.p_tr, lb.p_tr, rtol=1e-5, atol=1e-8)
    assert any(k[0] == 'trK' and not isinstance(v, str) for k, v in engs[0]._graphs.items())    # (the one-graph form did run)
    # a single step after K-step replays picks up its own schedule values again
    a.train.run(feed(a, 99)); b.train.run(feed(b, 99))
    torch.cuda.synchronize()
    assert rel(engs[0].P, engs[1].P) <= 1e-6


def test_k_steps_in_one_graph_with_the_input_pipeline():
    """run_steps with Dataset.bind_engine: launch 0 of step j gathers the batch staged in record slot j -- the same
    batches, in the same order, from the same numpy stream as step-by-step training."""
    import arch_and_hypers as A
    from lib.data import Dataset
    n, K = 32, 4
    outs = []
    for mode in ('single', 'steps'):
        ds = Dataset.synthetic(n_tr=300, n_ts=140, seed=1)
        net = A.ac_chain(k_cpt=1.6e-8, seed=5)(ds.x0_shape, ds.y_shape)
        eng = net.engine()
        x0, y = ds.bind_engine(eng, n)
        np.random.seed(21)
        f = lambda t: {net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: 0.02 / (1 + t), net.τ: 1.0}
        for call in range(3):
            if mode == 'single':
                for j in range(K):
                    ds.stage_training_draws(n, eng=eng)
                    net.train.run(f(call * K + j))
            else:
                if call == 1:                                  # (both ways of staging the K record slots)
                    for j in range(K):
                        ds.stage_training_draws(n, eng=eng, slot=j)
                else:
                    ds.stage_training_draws_k(K, n, eng=eng)
                net.train.run_steps([f(call * K + j) for j in range(K)])
        torch.cuda.synchronize()
        outs.append((eng.P.clone(), eng.x0[:n].clone(), np.random.get_state()[1].copy()))
    rel = lambda u, v: float((u - v).abs().max() / v.abs().max())
    assert torch.equal(outs[0][1], outs[1][1])                       # the last batch is the same batch
    assert np.array_equal(outs[0][2], outs[1][2])                    # ... and the numpy stream stands at the same place
    assert rel(outs[0][0], outs[1][0]) <= 1e-6
