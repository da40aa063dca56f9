"""GPU: the callers either side of the hot path (SURVEY 8 a16 and the 'next' rows): on-device
branch compaction, dataset-averaged statistics in the reference's schema, save/load, the
train-nets driver."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('n', [1, 63, 64, 65, 128, 1000, 1024, 1025, 5000])
def test_compact_by_branch(n):
    """Ordered indices of the samples that reach a node (p_ev > 0) and their count, on device:
    wave64 ballot + popcount prefix (empty, ragged and multi-pass sizes)."""
    from lib import _hip
    lib = _hip.load()
    rng = np.random.default_rng(n)
    for frac in (0.0, 0.3, 1.0):
        p = (rng.random(n) < frac).astype(np.float32)
        pd = torch.from_numpy(p).cuda()
        idx = torch.full((n,), -1, dtype=torch.int32, device='cuda')
        cnt = torch.zeros(1, dtype=torch.int32, device='cuda')
        _hip.check(lib.mpnn_compact_by_branch(pd.data_ptr(), n, idx.data_ptr(), cnt.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream), 'compact')
        want = np.nonzero(p > 0)[0]
        assert int(cnt.item()) == len(want)
        assert np.array_equal(idx.cpu().numpy()[:len(want)], want)


def make_trained_net(steps=6):
    import arch_and_hypers as A
    from lib.data import Dataset
    ds = Dataset.synthetic(n_tr=300, n_ts=140, seed=1)
    net = A.ac_chain(k_cpt=1.6e-8, seed=5)(ds.x0_shape, ds.y_shape)
    np.random.seed(0)
    for t in range(steps):
        x0, y = ds.augmented_training_batch(A.batch_size)
        net.train.run({net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: 0.05, net.τ: A.τ_ds(t)})
    return net, ds


def test_net_desc_schema_and_consistency():
    """desc.py:24-36 schema; ragged last batches (300 = 2*128 + 44); routing histogram sums to 1;
    acc = sum of per-leaf p_cor; moc between the first exit's cost and the full chain's."""
    from lib.desc import net_desc, render_net_desc
    net, ds = make_trained_net()
    desc = net_desc(net, ds, {net.τ: 1.0})
    assert set(desc) == {'type', 'stats_tr', 'stats_ts', 'root'} and desc['type'] == 'ActorNet'
    for key in ('stats_tr', 'stats_ts'):
        assert set(desc[key]) == {'acc', 'moc'}
        leaves, stack = [], [desc['root']]
        while stack:
            d = stack.pop()
            assert set(d) == {'name', 'stats_tr', 'stats_ts', 'sinks'}
            stack += d['sinks']
            if not d['sinks']:
                leaves.append(d)
        assert len(leaves) == 8
        for d in leaves:
            assert set(d[key]) == {'p_cor', 'p_inc', 'p_cor_by_cls', 'p_inc_by_cls', 'p_tr', 'c_err'}
            assert len(d[key]['p_cor_by_cls']) == 10
        hist = [d[key]['p_cor'] + d[key]['p_inc'] for d in leaves]
        assert abs(sum(hist) - 1) < 1e-6
        assert abs(sum(d[key]['p_cor'] for d in leaves) - desc[key]['acc']) < 1e-6
        assert 1368608 - 1 <= desc[key]['moc'] <= 20699872 + 1
        assert abs(sum(d[key]['p_tr'] for d in leaves) - 1) < 1e-4
    # the first switch carries x_rte (mean |router output|), train-nets:127-128
    first_block = desc['root']['sinks'][0]
    assert 'x_rte' in first_block['stats_ts']
    assert 'ActorNet' in render_net_desc(desc, 'demo')


def test_serdes_roundtrip(tmp_path):
    from lib.serdes import write_net, read_net
    net, ds = make_trained_net(3)
    path = str(tmp_path / 'net.npy')
    write_net(path, net, with_optimizer=True)
    rec = np.load(path, allow_pickle=True)[()]
    assert set(rec) == {'type', 'root', 'hypers', 'params'} and rec['type'] == 'ActorNet'      # serdes.py:40-44
    assert set(rec['root']) >= {'type', 'name', 'hypers', 'params', 'sinks', 'comps', 'router'}
    net2 = read_net(path)
    for p, q in zip(net._all_params, net2._all_params):
        assert (p.name, p.shape) == (q.name, q.shape) and torch.equal(p.data.cpu(), q.data.cpu())
        if p.trainable:
            assert torch.equal(p.accum.cpu(), q.accum.cpu())
    x0, y = next(ds.test_set())
    net.eval({net.x0: x0, net.y: y}); net2.eval({net2.x0: x0, net2.y: y})
    a, b = net.state(), net2.state()
    assert torch.equal(a[(net, 'moc')].cpu(), b[(net2, 'moc')].cpu())
    # and training continues identically from the restored momentum
    x0, y = next(ds.training_set())
    for m in (net, net2):
        m.train.run({m.x0: x0, m.y: y, m.mode: 'tr', m.λ_lrn: 0.05, m.τ: 1.0})
    assert np.abs(net.engine().P.cpu().numpy() - net2.engine().P.cpu().numpy()).max() < 1e-6


def test_train_nets_cli(tmp_path):
    out = str(tmp_path / 'nets')
    cmd = [sys.executable, os.path.join(ROOT, 'multipath-nn_amd', 'train-nets'), 'cifar10-cr', '--synthetic',
           '--iters', '4', '--log-every', '2', '--nets', '1', '--out', out]
    subprocess.check_call(cmd, cwd=str(tmp_path))
    base = os.path.join(out, 'cifar10-cr')
    for f in ('0001.npy', '0001-stats.npy', '0001-log.txt', '0001-stats/00000002.npy', '0001-stats/00000004.npy'):
        assert os.path.exists(os.path.join(base, f)), f                      # scripts/train-nets:149-157
    desc = np.load(os.path.join(base, '0001-stats.npy'), allow_pickle=True)[()]
    assert desc['type'] == 'CriticNet' and 0 <= desc['stats_ts']['acc'] <= 1


def test_train_adaptive_nets_cli(tmp_path):
    """train-adaptive-nets counterpart: per-sample k_cpt drawn each step, one stats file per k_cpt."""
    out = str(tmp_path / 'nets')
    cmd = [sys.executable, os.path.join(ROOT, 'multipath-nn_amd', 'train-adaptive-nets'), 'hybrid-ac-dynkcpt',
           '--synthetic', '--iters', '3', '--out', out]
    subprocess.check_call(cmd, cwd=str(tmp_path))
    base = os.path.join(out, 'hybrid-ac-dynkcpt')
    sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd'))
    import arch_and_hypers as A
    for i in range(len(A.k_cpts)):
        desc = np.load(os.path.join(base, '%.4i-stats.npy' % i), allow_pickle=True)[()]       # train-adaptive-nets:102-105
        assert desc['type'] == 'ActorNet' and 0 <= desc['stats_ts']['acc'] <= 1
    assert os.path.exists(os.path.join(base, 'net.npy'))


def test_device_augmentation_matches_reference_fixtures():
    """mpnn_augment_batch (dataset resident in HBM, draws from the host with the reference's RNG
    sequence) reproduces the batches the REFERENCE's scripts/lib/data.py produced for the same seeds
    (tests/golden/data_aug_golden.npz, generated by importing that module): gathered pixels exactly,
    the mean fill to fp32 rounding."""
    import os
    from lib import data as D
    gold = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'data_aug_golden.npz'))
    ds = D.Dataset(arrays=dict(x0_tr=gold['x0'], y_tr=gold['y'], x0_ts=gold['x0'][:1], y_ts=gold['y'][:1],
                               m_sym=gold['m_sym']))
    ds.to_device('cuda:0')
    for k in range(3):
        seed, n, r = (int(v) for v in gold['case%d_args' % k])
        np.random.seed(seed)
        x, y = ds.augmented_training_batch_device(n, r)
        torch.cuda.synchronize()
        want_x, want_y = gold['case%d_x' % k], gold['case%d_y' % k]
        assert np.array_equal(y.cpu().numpy(), want_y.astype(np.float32))
        err = np.abs(x.cpu().numpy().astype(np.float64) - want_x).max()
        assert err <= 1e-7, (k, err)
    # and the host path on the same seed draws the same batch (same RNG consumption)
    np.random.seed(123)
    xh, yh = ds.augmented_training_batch(16, 2)
    np.random.seed(123)
    xd, yd = ds.augmented_training_batch_device(16, 2)
    assert np.abs(xd.cpu().numpy() - xh.astype(np.float32)).max() <= 1e-7
    assert np.array_equal(yd.cpu().numpy(), yh.astype(np.float32))


def _resolve(obj, keys):
    for k in keys:
        obj = obj[k]
    return obj


def test_files_written_by_the_cli_serve_the_reference_consumers(tmp_path):
    """SURVEY 8 f3.  tests/golden/access_paths.json holds every dictionary access path the
    reference's figure scripts (scripts/make-routing-hists:14-28, make-acc-eff-plots:25-27,
    make-nlds, make-pres-figs, make-videos) and its loader (scripts/lib/serdes.py:21-60) take into
    `<i>-stats.npy`, `<i>-stats/<t>.npy` and `<i>.npy` (extracted from the reference tree by
    tests/golden/make_access_paths.py).  Every path must resolve on the files THIS build's
    train-nets writes -- both with the dense and with the routed statistics pass -- and the
    routing histogram the reference computes from them must be a distribution."""
    import json
    paths = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'access_paths.json'), encoding='utf-8'))
    for extra in ([], ['--routed-stats', '--stats-batch', '100']):
        out = str(tmp_path / ('nets' + str(len(extra))))
        cmd = [sys.executable, os.path.join(ROOT, 'multipath-nn_amd', 'train-nets'), 'cifar10-ac', '--synthetic',
               '--iters', '2', '--log-every', '2', '--nets', '3', '--out', out] + extra
        subprocess.check_call(cmd, cwd=str(tmp_path))
        base = os.path.join(out, 'cifar10-ac')
        files = [os.path.join(base, '0003-stats.npy'), os.path.join(base, '0003-stats', '00000002.npy')]
        for f in files:
            log = np.load(f, allow_pickle=True)[()]                  # the consumers' np.load(p)[()]
            blocks = [_resolve(log, ['root', 'sinks', 0])]
            while len(blocks[-1]['sinks']) > 1:                      # their `ℓ = ℓ['sinks'][1]` walk down the chain
                blocks.append(blocks[-1]['sinks'][1])
            assert len(blocks) == 8
            for script, chains in paths['stats'].items():
                for c in chains:
                    if c['var'] in ('net', 'log'):
                        _resolve(log, c['keys'])
                    else:                                            # a block descriptor
                        for b in blocks[:-1] if c['keys'] == ['sinks', 1] else blocks:
                            v = _resolve(b, c['keys'])
                            if c['keys'][-1].__class__ is str and c['keys'][-1].endswith('_by_cls'):
                                assert len(v) == 10
            # get_p_ev of make-routing-hists:14-28
            p_ev = [b['sinks'][0]['stats_ts']['p_cor'] + b['sinks'][0]['stats_ts']['p_inc'] for b in blocks]
            assert abs(sum(p_ev) - 1) < 1e-6 and min(p_ev) >= 0
            assert 0 <= log['stats_ts']['acc'] <= 1 and 1368608 - 1 <= log['stats_ts']['moc'] <= 20699872 + 1
        # the saved net: every key scripts/lib/serdes.py reads back, on every layer record
        rec = np.load(os.path.join(base, '0003.npy'), allow_pickle=True)[()]
        layer_keys = {c['keys'][0] for c in paths['net']} - {'root'}
        assert {'type', 'root', 'hypers', 'params'} <= set(rec)
        stack, n_layers = [rec['root']], 0
        while stack:
            r = stack.pop()
            if r is None:
                continue
            n_layers += 1
            assert layer_keys <= set(r), (layer_keys - set(r))
            assert all(isinstance(v, np.ndarray) for v in r['params'].values())
            stack += list(r['sinks']) + list(r['comps']) + [r['router']]
        assert n_layers > 100


def test_tree_experiments_run_from_the_cli(tmp_path):
    """`hybrid-ac-tree` (scripts/train-nets:52-54) and its adaptive counterpart: the 47-block tree through
    both drivers -- training steps, the statistics pass, the saved net."""
    out = str(tmp_path / 'nets')
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'multipath-nn_amd', 'train-nets'), 'hybrid-ac-tree', '--synthetic',
                           '--iters', '2', '--log-every', '2', '--nets', '2', '--out', out, '--routed-stats'], cwd=str(tmp_path))
    desc = np.load(os.path.join(out, 'hybrid-ac-tree', '0002-stats.npy'), allow_pickle=True)[()]
    assert desc['type'] == 'ActorNet' and len(desc['root']['sinks'][0]['sinks']) == 3        # exit + two sub-trees
    leaves, stack = [], [desc['root']]
    while stack:
        d = stack.pop()
        stack += d['sinks']
        if not d['sinks']:
            leaves.append(d)
    assert len(leaves) == 47
    assert abs(sum(d['stats_ts']['p_cor'] + d['stats_ts']['p_inc'] for d in leaves) - 1) < 1e-6
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'multipath-nn_amd', 'train-adaptive-nets'), 'hybrid-cr-tree-dynkcpt',
                           '--synthetic', '--iters', '2', '--out', out], cwd=str(tmp_path))
    assert os.path.exists(os.path.join(out, 'hybrid-cr-tree-dynkcpt', '0007-stats.npy'))


@pytest.mark.parametrize('form', ['one_graph', 'sections', 'eager'])
def test_bound_input_pipeline_feeds_every_step_in_every_graph_form(form):
    """Dataset.bind_engine makes mpnn_augment_batch launch 0 of the step.  In EVERY form of the step -- one hipGraph,
    one graph per gradient-bucket section with host-issued collectives (the data-parallel fallback), eager launches -- a
    step must train on the batch staged for it (round 4's section-graph form replayed without the gather: every step
    after the warm-up trained on the first batch), also after the engine reallocated its input buffers for a larger
    evaluation batch (a statistics pass at 4 096 images between two steps), and binding must not consume numpy draws."""
    import arch_and_hypers as A
    from lib.data import Dataset
    ds = Dataset.synthetic(n_tr=300, n_ts=140, seed=1)
    net = A.ac_chain(k_cpt=1.6e-8, seed=5)(ds.x0_shape, ds.y_shape)
    eng = net.engine()
    n = 32
    if form == 'eager':
        eng.use_graph = False
    if form == 'sections':
        calls = []

        def stub(flat):                   # a collective that cannot be captured (gloo / host-issued): world of one
            calls.append(flat.numel())
            return None
        eng.world, eng.allreduce, eng.allreduce_capturable = 1, stub, False
        eng._graphs.clear()
    np.random.seed(3)
    state = np.random.get_state()[1].copy()
    x0, y = ds.bind_engine(eng, n)
    assert np.array_equal(np.random.get_state()[1], state), 'bind_engine consumed draws of the global numpy stream'
    ref = Dataset.synthetic(n_tr=300, n_ts=140, seed=1)
    np.random.seed(3)
    seen = []
    for t in range(6):
        ds.stage_training_draws(n, eng=eng)
        net.train.run({net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: 0.01, net.τ: 1.0})
        torch.cuda.synchronize()
        got = eng.x0[:n].cpu().numpy().copy()
        rng_state = np.random.get_state()
        np.random.seed(3)                                      # the same draws through the host path, step t
        for _ in range(t + 1):
            want, want_y = ref.augmented_training_batch(n)
        np.random.set_state(rng_state)
        assert np.abs(got - want).max() <= 1e-6, (form, t)
        assert np.array_equal(eng.y[:n].cpu().numpy(), want_y)
        seen.append(got)
        if t == 3:                                             # a larger evaluation batch: the buffers are reallocated
            xb = torch.rand(n * 8, *ds.x0_shape, device='cuda')
            yb = torch.zeros(n * 8, ds.y_shape[0], device='cuda'); yb[:, 0] = 1
            net.eval({net.x0: xb, net.y: yb})
    assert all(np.abs(a - b).max() > 0 for a, b in zip(seen, seen[1:]))
    # the old call form -- staging into the dataset's own buffer while an engine is bound -- would feed the step from records
    # nobody wrote (every sample source image 0, unflipped, unshifted): it raises instead
    with pytest.raises(ValueError):
        ds.stage_training_draws(n)
    if form == 'sections':
        assert len(calls) >= 6 and eng._graphs and all(v[1] != 'whole' for v in eng._graphs.values() if isinstance(v, tuple))


def test_adaptive_loop_draws_in_the_reference_order():
    """train-adaptive-nets with several iterations per hipGraph replay: per iteration the reference draws the batch's
    augmentation records and THEN rand.choice(k_cpts, batch) (scripts/train-adaptive-nets:24-27,93-97).  Staging K
    iterations at once keeps that order (stage_training_draws_k(between=...)): records and k_cpt vectors equal the
    per-iteration loop's on the same seed, and so does the stream position afterwards."""
    import arch_and_hypers as A
    from lib import data as D
    ds = D.Dataset.synthetic(n_tr=300, n_ts=140, seed=1)
    ds.m_sym = np.array([1, 0, 1, 1, 0, 0, 1, 0, 1, 1], bool)
    ds.to_device('cuda:0')
    n, K = 24, 4
    np.random.seed(5)
    want = []
    for t in range(2 * K):
        j, flip, sh = D._draw_augmentation(n, 300, ds.y_tr, ds.m_sym, 4)
        want.append((j, flip, sh, np.random.choice(A.k_cpts, n)))
    tail = np.random.randint(0, 2 ** 32, 8, dtype=np.uint32)
    np.random.seed(5)
    got_k = []
    for call in range(2):
        ks = [None] * K

        def between(jj):
            ks[jj] = np.random.choice(A.k_cpts, n)
        ds.stage_training_draws_k(K, n, between=between)
        torch.cuda.synchronize()
        rec = ds._draw_buffers(n)['dev'][:K, :n].cpu().numpy()
        for jj in range(K):
            j, flip, sh, kc = want[call * K + jj]
            assert np.array_equal(rec[jj][:, 0], j) and np.array_equal(rec[jj][:, 1].astype(bool), flip) and np.array_equal(rec[jj][:, 2:], sh)
            assert np.array_equal(ks[jj], kc)
    assert np.array_equal(np.random.randint(0, 2 ** 32, 8, dtype=np.uint32), tail)


@pytest.mark.parametrize('kind', ['cr_tree', 'sr', 'dyn', 'wide', 'conv', 'random_tree'])
def test_checkpoint_round_trip_for_every_net_family(kind, tmp_path):
    """write_net / read_net (the reference's record format + the momentum slots) on the other net families: a 47-block
    critic tree, a statically-routed chain, a per-sample-k_cpt net, wide routers with 37 classes (the any-width exit
    kernels), a single-scale Conv net, a randomly drawn tree with static links -- the restored net has the same
    parameters, momentum and BatchNorm state and takes the same next training step, bit for bit."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import arch_and_hypers as A
    from lib.net_types import CriticNet
    from lib.serdes import write_net, read_net
    from test_net_parity import _wide_chain, batch, perturb_routers
    hw, n_cls, extra = 32, 10, (lambda net: {net.τ: 0.6})
    if kind == 'cr_tree':
        mk = A.cr_tree(k_cpt=4e-9)
    elif kind == 'sr':
        mk, extra = A.sr_chain(4), (lambda net: {})
    elif kind == 'dyn':
        mk = A.ac_chain(dyn_k_cpt=True)
        extra = lambda net: {net.τ: 0.6, net.k_cpt: np.random.default_rng(1).choice(A.k_cpts, 12).astype(np.float32)}
    elif kind == 'wide':
        mk, n_cls = _wide_chain(CriticNet, (24, 40), k_cpt=8e-9), 37
    elif kind == 'conv':
        from test_conv_layer import pooled_conv_net
        mk, hw, extra = pooled_conv_net(), 16, (lambda net: {})
    else:
        from lib.net_types import ActorNet
        from test_fuzz_trees_gpu import draw_tree, make_tree
        mk = make_tree(ActorNet, draw_tree(np.random.default_rng(1300), 3), k_cpt=4e-9)
    net = mk((hw, hw, 3), (n_cls,))
    net.engine().init_params(21)
    if kind not in ('sr', 'conv'):
        perturb_routers(net)
    x0, y = batch(12, 3, n_cls, seed=1, hw=hw)
    feed = lambda m: {m.x0: x0, m.y: y, m.mode: 'tr', m.λ_lrn: 0.05, **extra(m)}
    for _ in range(2):
        net.train.run(feed(net))
    path = str(tmp_path / 'net.npy')
    write_net(path, net, with_optimizer=True)
    net2 = read_net(path)
    assert type(net2) is type(net)
    for p, q in zip(net._all_params, net2._all_params):
        assert (p.name, p.shape) == (q.name, q.shape) and torch.equal(p.data.cpu(), q.data.cpu()), (p.owner.name, p.name)
        if p.trainable:
            assert torch.equal(p.accum.cpu(), q.accum.cpu()), (p.owner.name, p.name)
    net.train.run(feed(net)); net2.train.run(feed(net2))
    torch.cuda.synchronize()
    tol = 0.0 if kind != 'conv' else 2e-6         # (the Conv engine's 1x1 weight gradients meet in fp32 atomics)
    d = float((net.engine().P - net2.engine().P).abs().max())
    assert d <= tol * float(net.engine().P.abs().max()), d


@pytest.mark.parametrize('seed', [0, 1, 2, 3, 4, 5])
def test_device_augmentation_on_random_shapes(seed):
    """mpnn_augment_batch / mpnn_augment_batch_multi against the vectorised host path (itself pinned to the reference's
    fixtures above) on random image sizes (8 ... 40 pixels, non-square too), 1 / 3 / 4 channels, 2 ... 12 classes, shift
    ranges 0 ... 6 (beyond half of a small image), ragged batches: gathered pixels exact, mean fill to fp32 rounding."""
    import ctypes as C
    from lib import _hip, data as D
    rng = np.random.default_rng(500 + seed)
    H, W = int(rng.integers(8, 41)), int(rng.integers(8, 41))
    if seed % 2 == 0:
        W = H
    c, n_cls, n_src = int(rng.choice([1, 3, 4])), int(rng.integers(2, 13)), int(rng.integers(1, 70))
    x = rng.random((n_src, H, W, c)).astype(np.float32)
    yy = np.eye(n_cls, dtype=np.float32)[rng.integers(0, n_cls, n_src)]
    ds = D.Dataset(arrays=dict(x0_tr=x, y_tr=yy, x0_ts=x[:1], y_ts=yy[:1], m_sym=rng.random(n_cls) < 0.5))
    ds.to_device('cuda:0')
    for trial in range(3):
        n, r = int(rng.integers(1, 150)), int(rng.integers(0, 7))
        np.random.seed(seed * 10 + trial)
        xh, yh = ds.augmented_training_batch(n, r)
        np.random.seed(seed * 10 + trial)
        xd, yd = ds.augmented_training_batch_device(n, r)
        torch.cuda.synchronize()
        assert np.abs(xd.cpu().numpy().astype(np.float64) - xh).max() <= 2e-7, (H, W, c, n, r)
        assert np.array_equal(yd.cpu().numpy(), yh.astype(np.float32))
    # the multi-consumer launch (co-training): three consumers, each its own records
    n, r, K = 17, 3, 3
    np.random.seed(99)
    want = [ds.augmented_training_batch(n, r) for _ in range(K)]
    np.random.seed(99)
    recs = torch.zeros((K, n, 4), dtype=torch.int32)
    for k in range(K):
        D._draw_augmentation_fast(n, n_src, ds._sym_u8, r, out=recs[k].numpy(), all_sym=ds._all_sym)
    recs_d = recs.cuda()
    xo = torch.full((K, n, H, W, c), float('nan'), device='cuda'); yo = torch.full((K, n, n_cls), float('nan'), device='cuda')
    tab = []
    for k in range(K):
        d = _hip.AugmentDst()
        d.draw, d.x_out, d.y_out = recs_d[k].data_ptr(), xo[k].data_ptr(), yo[k].data_ptr()
        tab.append(d)
    dev_tab = _hip.to_device_table(tab, 'cuda:0')
    _hip.check(_hip.load().mpnn_augment_batch_multi(ds._x_dev.data_ptr(), ds._y_dev.data_ptr(), dev_tab.data_ptr(), K, n, H, W, c, n_cls,
                                                    torch.cuda.current_stream().cuda_stream), 'augment_batch_multi')
    torch.cuda.synchronize()
    for k in range(K):
        assert np.abs(xo[k].cpu().numpy().astype(np.float64) - want[k][0]).max() <= 2e-7, k
        assert np.array_equal(yo[k].cpu().numpy(), want[k][1].astype(np.float32)), k


def test_engine_refuses_conv_batchnorms_with_different_decays():
    """The conv BatchNorms' moving-average decay is ONE kernel argument per net.  Trees built through the layer classes always
    satisfy that (MultiscaleBatchNorm hands every scale a default BatchNorm, as the reference does); a tree whose components
    were edited by hand is refused instead of trained with block 0's number (round 5: silently)."""
    import arch_and_hypers as A
    net = A.ac_chain(k_cpt=1e-9, seed=3)((32, 32, 3), (10,))
    blk = [ℓ for ℓ in net.layers if ℓ.name == 'ReConvMax'][1]
    blk.comps[1].comps[0].hypers.d = 0.5
    with pytest.raises(NotImplementedError):
        net.engine()


@pytest.mark.parametrize('kind', ['ac', 'cr', 'sr', 'tree'])
def test_state_sums_equal_the_sums_of_state(kind):
    """Engine.state_sums (what the statistics pass accumulates: a dozen batched device operations) == the per-sample
    statistics of Engine.state summed over the batch, key for key -- dense and routed evaluation, chains, a tree with 3-way
    switches (x_rte averages over the node's own number of sinks)."""
    import arch_and_hypers as A
    mk = {'ac': A.ac_chain(k_cpt=1e-9, seed=3), 'cr': A.cr_chain(k_cpt=1e-9, seed=3), 'sr': A.sr_chain(5), 'tree': A.ac_tree(k_cpt=1e-9, seed=3)}[kind]
    net = mk((32, 32, 3), (10,))
    eng = net.engine()
    rng = np.random.default_rng(0)
    if kind != 'sr':
        for ℓ in net.layers:
            if ℓ.router is not None:
                w = ℓ.router.comps[-1].params.w
                w.assign(rng.standard_normal(w.shape) * 0.5)
    n = 300 if kind != 'tree' else 40
    x0 = rng.random((n, 32, 32, 3)).astype(np.float32)
    y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, n)]
    for routed in ([False, True] if kind != 'sr' else [False]):
        net.eval({net.x0: x0, net.y: y}, routed=routed)
        want = {k: v.double().sum(0) for k, v in eng.state().items()}
        got = eng.state_sums()
        assert set(got) == set(want)
        for k in want:
            assert got[k].shape == want[k].shape, k[1]
            assert torch.allclose(got[k], want[k], rtol=1e-6, atol=1e-6), (kind, routed, k[1])


def test_train_nets_cli_with_a_second_experiment(tmp_path):
    """`train-nets cifar10-ac --with cifar10-cr`: the nets of two experiments on one dataset advance together (the -ac and -cr
    chains share an architecture: they share launches).  Every experiment keeps its own files, net types and hyper-parameter
    schedule."""
    out = str(tmp_path / 'both')
    cmd = [sys.executable, os.path.join(ROOT, 'multipath-nn_amd', 'train-nets'), 'cifar10-ac', '--with', 'cifar10-cr',
           '--synthetic', '--iters', '6', '--log-every', '3', '--nets', '0', '3', '--out', out]
    res = subprocess.run(cmd, cwd=str(tmp_path), capture_output=True)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    assert b'co-training 4 nets' in res.stdout
    for ex, kind in (('cifar10-ac', 'ActorNet'), ('cifar10-cr', 'CriticNet')):
        for i in (0, 3):
            for f in ('%.4i.npy' % i, '%.4i-stats.npy' % i, '%.4i-log.txt' % i, '%.4i-stats/00000003.npy' % i, '%.4i-stats/00000006.npy' % i):
                assert os.path.exists(os.path.join(out, ex, f)), (ex, f)
            d = np.load(os.path.join(out, ex, '%.4i-stats.npy' % i), allow_pickle=True)[()]
            assert d['type'] == kind and 0 <= d['stats_ts']['acc'] <= 1 and np.isfinite(d['stats_tr']['moc'])
