"""Routed evaluation: on-device per-branch compaction in 'ev' mode.

The reference evaluates every block densely and multiplies the 0/1 masks p_ev into the statistics
(scripts/lib/net_types.py:127-131, scripts/train-nets:117-130).  The routed program runs a block
only on the samples its ancestors' routers sent to it.  Checked here, through the C ABI:

  * routed == dense, EXACTLY (same kernels, per-sample arithmetic independent of the slot): p_ev of
    every node, and c_err / delta_cor / router.x wherever the sample reaches the node (0 elsewhere);
    hence acc, moc, p_cor, p_inc, p_*_by_cls are bit-identical;
  * both against the float64 oracle: per-sample c_err 2e-4, p_ev exact, acc / moc / routing
    histogram within 1e-3 (north_star);
  * the device-side sample lists: sorted(list) == nonzero(p_ev(block)), counts on the device;
  * blocks nobody is routed to are not executed (their buffers keep a NaN poison);
  * router states: all-exit-0 (initialisation, KA3), exit fractions [1/8] x 8, random;
    batch sizes 64 (oracle), 1000 (ragged: not a multiple of the 16-sample / 4-image tiles), 4096.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def batch(n, seed=0):
    rng = np.random.default_rng(seed)
    x0 = rng.random((n, 32, 32, 3)).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, n)]
    return x0, y


def make(kind='ac', seed=7, **hyp):
    import arch_and_hypers as A
    net = (A.ac_chain if kind == 'ac' else A.cr_chain)(**hyp)((32, 32, 3), (10,))
    eng = net.engine()
    eng.init_params(seed)
    # moving averages away from their (0, 1) initial values, as after training
    rng = np.random.default_rng(seed + 1)
    for p in net._all_params:
        if not p.trainable:
            v = rng.random(p.shape) * 0.5 + (0.75 if p.name == 'v_avg' else -0.25)
            p.assign(v)
    return net


def randomise_routers(net, seed=5, scale=0.5):
    rng = np.random.default_rng(seed)
    for ℓ in net.layers:
        if ℓ.router is not None:
            last = ℓ.router.comps[-1].params
            last.w.assign(rng.standard_normal(last.w.shape) * scale)
            last.b.assign(rng.standard_normal(last.b.shape) * 0.1)


def calibrate_exit_fractions(net, x0, y, fractions):
    """Shift the routers' exit-sink biases so that `fractions[k]` of the batch leaves at exit k
    (chain nets: sink 0 = exit, sink 1 = continue)."""
    net.eval({net.x0: x0, net.y: y})
    n = len(x0)
    alive = np.ones(n, bool)
    for k, ℓ in enumerate(net.switches):
        r = ℓ.router.x.cpu().numpy().astype(np.float64)
        want = int(round(fractions[k] * n))
        margin = r[:, 1] - r[:, 0]                    # exits iff r0 + shift >= r1  (arg-max, first index on ties)
        idx = np.flatnonzero(alive)
        order = idx[np.argsort(margin[idx], kind='stable')]
        take = order[:want]
        if want == 0:
            shift = margin[idx].min() - 1.0 if len(idx) else 0.0
        elif want >= len(idx):
            shift = margin[idx].max() + 1.0 if len(idx) else 0.0
        else:
            shift = 0.5 * (margin[order[want - 1]] + margin[order[want]])
        b = ℓ.router.comps[-1].params.b
        v = b.numpy().copy()
        v[0] += shift
        b.assign(v)
        alive[take] = False


def snapshot(net):
    eng = net.engine()
    out = {'p_ev': {}, 'c_err': {}, 'd_cor': {}, 'r': {}}
    for nd in eng.nodes:
        out['p_ev'][nd.idx] = nd.layer.p_ev.cpu().numpy().copy()
    for nd in eng.leaves:
        out['c_err'][nd.idx] = nd.layer.c_err.cpu().numpy().copy()
        out['d_cor'][nd.idx] = nd.layer.δ_cor.cpu().numpy().copy()
    for nd in eng.switches:
        out['r'][nd.idx] = nd.layer.router.x.cpu().numpy().copy()
    st = net.state()
    out['state'] = {(k[1], getattr(k[0], 'name', 'net'), id(k[0])): v.cpu().numpy().copy() for k, v in st.items()}
    return out


def check_routed_equals_dense(net, x0, y, feed_extra=None, modes=(True, 1, 3)):
    """modes: routed=True (the engine picks the depth from which blocks gather by batch size), 1 (every block below the
    root gathers: the work-minimal program), 3 (the convs of blocks 0-2 run on every sample, deeper blocks gather)."""
    dense = None
    for mode in modes:
        dense = _check_routed_equals_dense(net, x0, y, feed_extra, mode)
    return dense


def _check_routed_equals_dense(net, x0, y, feed_extra, mode):
    eng = net.engine()
    feed = {net.x0: x0, net.y: y, **(feed_extra or {})}
    net.eval(feed)
    torch.cuda.synchronize()
    dense = snapshot(net)
    net.eval(feed, routed=mode)
    torch.cuda.synchronize()
    routed = snapshot(net)
    parent_leaf = {nd.idx: nd.parent for nd in eng.leaves}
    for k in dense['p_ev']:
        assert np.array_equal(dense['p_ev'][k], routed['p_ev'][k]), ('p_ev', k)
    for k in dense['c_err']:
        reach = dense['p_ev'][parent_leaf[k]] > 0            # the samples that reach the leaf's block
        assert np.array_equal(dense['c_err'][k][reach], routed['c_err'][k][reach]), ('c_err', k)
        assert np.array_equal(dense['d_cor'][k][reach], routed['d_cor'][k][reach]), ('d_cor', k)
        assert not routed['c_err'][k][~reach].any() and not routed['d_cor'][k][~reach].any()
    for k in dense['r']:
        reach = dense['p_ev'][k] > 0
        assert np.array_equal(dense['r'][k][reach], routed['r'][k][reach]), ('router.x', k)
        assert not routed['r'][k][~reach].any()
    for key, v in dense['state'].items():
        if key[0] in ('acc', 'moc', 'p_cor', 'p_inc', 'p_cor_by_cls', 'p_inc_by_cls'):
            assert np.array_equal(v, routed['state'][key]), key
    # the sample lists the routers wrote on the device
    for b in eng.blocks:
        if b.ev_list is None:
            continue
        cnt = int(b.ev_list[1].cpu()[0])
        got = np.sort(b.ev_list[0][:cnt].cpu().numpy())
        want = np.flatnonzero(dense['p_ev'][b.node.idx] > 0)
        assert cnt == len(want) and np.array_equal(got, want), ('list', b.node.idx, cnt, len(want))
    return dense


def check_vs_oracle(net, x0, y, dense):
    from oracle.ref_net import RefNet
    eng = net.engine()
    ref = RefNet(net)
    ref.load_params()
    res = ref.forward(x0, y, 'ev')
    R = lambda ℓ: res['out'][id(ℓ)]
    for nd in eng.nodes:
        assert np.array_equal(dense['p_ev'][nd.idx], R(nd.layer)['p_ev'].numpy()), ('p_ev vs oracle', nd.idx)
    for nd in eng.leaves:
        ce = R(nd.layer)['c_err'].detach().numpy()
        assert np.abs(dense['c_err'][nd.idx] - ce).max() < 2e-4 * (1 + np.abs(ce).max())
        assert np.array_equal(dense['d_cor'][nd.idx], R(nd.layer)['δ_cor'].numpy())
    for nd in eng.switches:
        rx = R(nd.layer.router)['x'].detach().numpy()
        assert np.abs(dense['r'][nd.idx] - rx).max() < 2e-4 * (1 + np.abs(rx).max())
    rs = ref.stats(res)
    st = {k[0]: v for k, v in dense['state'].items() if k[1] == 'net'}
    assert abs(st['acc'].mean() - rs['acc'].mean()) <= 1e-3
    assert abs(st['moc'].mean() - rs['moc'].mean()) <= 1e-3 * rs['moc'].mean()
    hist = np.stack([dense['p_ev'][nd.idx] for nd in eng.leaves]).mean(1)
    assert np.abs(hist - rs['p_leaf'].mean(1)).max() <= 1e-3
    return hist


def test_all_exit_0_at_initialisation():
    """KA3: the last router map starts at zero -> everything leaves at exit 0; deeper blocks never run."""
    net = make()
    eng = net.engine()
    x0, y = batch(64)
    eng._ensure_capacity(64)
    for b in eng.blocks[1:]:
        for t in b.s:
            t.fill_(float('nan'))
    dense_feed = {net.x0: x0, net.y: y}
    net.eval(dense_feed, routed=1)           # (every block below the root gathers: nothing else may run)
    torch.cuda.synchronize()
    hist = [float(nd.layer.p_ev.mean()) for nd in eng.leaves]
    assert hist == [1.0] + [0.0] * 7
    st = net.state()
    assert float(st[(net, 'moc')].mean()) == 1361664 + 4384 + 2560
    assert np.isfinite(st[(net, 'acc')].cpu().numpy()).all()
    for b in eng.blocks[1:]:                 # not executed: the poison is still there
        assert torch.isnan(b.s[-1][:64]).all()
    dense = check_routed_equals_dense(net, x0, y)
    check_vs_oracle(net, x0, y, dense)


@pytest.mark.parametrize('kind', ['ac', 'cr'])
def test_exit_fractions_one_eighth_each(kind):
    net = make(kind, k_cpt=1e-9)
    randomise_routers(net)
    x0, y = batch(64, seed=3)
    calibrate_exit_fractions(net, x0, y, [1 / 8] * 7)
    dense = check_routed_equals_dense(net, x0, y)
    hist = check_vs_oracle(net, x0, y, dense)
    assert np.allclose(hist, 1 / 8), hist


@pytest.mark.parametrize('kind', ['ac', 'cr'])
def test_prefix_walk_at_every_depth(kind):
    """The dense prefix's exits in ONE launch + mpnn_ev_prefix_walk (d0 >= 2: lib/_plan.py:_program_ev): for every prefix
    depth of the 8-block chain -- up to d0 = 8, where nothing is left to gather and the walk only clears the entries of
    the samples that do not reach a node -- the routed pass still equals the dense one exactly, unreached entries are 0
    and the frontier's list == nonzero(p_ev); and it equals the exit-by-exit routed pass (MPNN_EV_PREFIX_WALK=0)."""
    import os
    net = make(kind, seed=21, k_cpt=1e-9)
    randomise_routers(net, seed=4, scale=1.0)
    x0, y = batch(200, seed=9)
    calibrate_exit_fractions(net, x0, y, [0.1, 0.2, 0.05, 0.15, 0.1, 0.1, 0.1])
    check_routed_equals_dense(net, x0, y, modes=(2, 4, 5, 7, 8))
    eng = net.engine()
    net.eval({net.x0: x0, net.y: y}, routed=5)
    torch.cuda.synchronize()
    assert any(op.what == 'ev_prefix_walk' for op in eng.program('ev', 200, routed=5)['fwd'])
    with_walk = snapshot(net)
    os.environ['MPNN_EV_PREFIX_WALK'] = '0'
    try:
        eng._progs.clear(); eng._graphs.clear()
        net.eval({net.x0: x0, net.y: y}, routed=5)
        torch.cuda.synchronize()
        assert not any(op.what == 'ev_prefix_walk' for op in eng.program('ev', 200, routed=5)['fwd'])
        without = snapshot(net)
    finally:
        del os.environ['MPNN_EV_PREFIX_WALK']
        eng._progs.clear(); eng._graphs.clear()
    for grp in ('p_ev', 'c_err', 'd_cor', 'r'):
        for k in with_walk[grp]:
            assert np.array_equal(with_walk[grp][k], without[grp][k]), (grp, k)


def test_tree_net_routed_equals_dense_at_600():
    """A TREE (ac_tree: multi-child blocks, 3-way switches) at an evaluation-size batch: the strip bodies and the
    32-channel tiles run inside the tree's groups too; routed == dense bit for bit, sample lists == nonzero(p_ev)."""
    import arch_and_hypers as A
    net = A.ac_tree(k_cpt=1e-9)((32, 32, 3), (10,))
    eng = net.engine()
    eng.init_params(13)
    rng = np.random.default_rng(14)
    for p in net._all_params:
        if not p.trainable:
            p.assign(rng.random(p.shape) * 0.5 + (0.75 if p.name == 'v_avg' else -0.25))
    randomise_routers(net, seed=3, scale=1.0)
    x0, y = batch(600, seed=12)
    dense = check_routed_equals_dense(net, x0, y, modes=(True, 1, 2, 3, 4))
    hist = np.stack([dense['p_ev'][nd.idx] for nd in eng.leaves]).mean(1)
    assert (hist > 0).sum() >= 4, hist


def test_random_routers_ragged_1000():
    """1000 samples: 62.5 sixteen-sample tiles, 250 four-image tiles; sub-batches of every size."""
    net = make(seed=11)
    randomise_routers(net, seed=9, scale=1.0)
    x0, y = batch(1000, seed=5)
    dense = check_routed_equals_dense(net, x0, y)
    hist = np.stack([dense['p_ev'][nd.idx] for nd in net.engine().leaves]).mean(1)
    assert (hist > 0).sum() >= 4, hist                 # a real spread over the exits
    # the first 48 samples against the oracle (per-sample independence: a sub-batch gives the same rows)
    net.eval({net.x0: x0[:48], net.y: y[:48]})
    sub = snapshot(net)
    for k in sub['p_ev']:
        assert np.array_equal(sub['p_ev'][k], dense['p_ev'][k][:48])
    check_vs_oracle(net, x0[:48], y[:48], sub)


def test_dense_eval_beyond_128_matches_small_batches():
    """The 'ev' exit path has no 128-sample cap: one 300-sample pass == three 100-sample passes."""
    net = make(seed=21)
    randomise_routers(net, seed=2)
    x0, y = batch(300, seed=8)
    net.eval({net.x0: x0, net.y: y})
    big = snapshot(net)
    for c in range(3):
        sl = slice(100 * c, 100 * c + 100)
        net.eval({net.x0: x0[sl], net.y: y[sl]})
        small = snapshot(net)
        for k in small['p_ev']:
            assert np.array_equal(small['p_ev'][k], big['p_ev'][k][sl])
        for k in small['c_err']:
            assert np.array_equal(small['c_err'][k], big['c_err'][k][sl])


def test_routed_4096_with_dyn_k_cpt():
    import arch_and_hypers as A
    net = A.ac_chain(dyn_k_cpt=True)((32, 32, 3), (10,))
    net.engine().init_params(3)
    randomise_routers(net, seed=4, scale=1.0)
    x0, y = batch(4096, seed=6)
    kc = np.random.default_rng(1).choice(A.k_cpts, 4096).astype(np.float32)
    check_routed_equals_dense(net, x0, y, {net.k_cpt: kc})


def test_hipgraph_replay_of_the_routed_program():
    """Counts live on the device: the captured graph replays on NEW data without re-capture."""
    net = make(seed=31)
    randomise_routers(net, seed=3, scale=1.0)
    eng = net.engine()
    feeds = [batch(256, seed=s) for s in (1, 2, 3, 4)]
    want = []
    eng.use_graph = False
    for x0, y in feeds:
        net.eval({net.x0: x0, net.y: y}, routed=True)
        want.append(snapshot(net))
    eng.use_graph = True
    for rep in range(2):
        for (x0, y), w in zip(feeds, want):
            net.eval({net.x0: x0, net.y: y}, routed=True)
            got = snapshot(net)
            for k in w['p_ev']:
                assert np.array_equal(w['p_ev'][k], got['p_ev'][k])
            for k in w['c_err']:
                assert np.array_equal(w['c_err'][k], got['c_err'][k])
