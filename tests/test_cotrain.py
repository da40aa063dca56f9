"""GPU: co-training (lib/_co.py) -- the nets of one experiment advance together, one launch per layer for all of them
(`mpnn_msconv_fwd_group_rep`, `mpnn_msconv_bwd_level_rep`, `mpnn_route_multi`, `mpnn_backward_finish_opt_multi`).  Each
net keeps the reference's semantics: its own batch, BatchNorm statistics, parameters.  Checked against (a) the same net
stepped ALONE from the same state and (b) the float64 oracle, decision-forced, at the whole-net tolerances."""
import numpy as np
import pytest
import torch

from test_net_parity import batch, perturb_routers, run_case

pytestmark = pytest.mark.gpu


def _nets(makers, seed0=100):
    nets = []
    for i, mk in enumerate(makers):
        net = mk((32, 32, 3), (10,))
        net.engine().init_params(seed0 + i)
        if net._net_kind != 'sr':
            perturb_routers(net, seed=5 + i)
        nets.append(net)
    return nets


def _copy_state(src, dst):
    es, ed = src.engine(), dst.engine()
    for a, b in ((es.P, ed.P), (es.A, ed.A), (es.S, ed.S)):
        b.copy_(a)
    ed.invalidate_packs()


@pytest.mark.parametrize('kind,K,n', [('ac', 3, 32), ('cr', 2, 16), ('ac', 8, 128), ('sr', 3, 16), ('tree', 2, 8)])
def test_cotrained_step_equals_the_solo_step(kind, K, n):
    """Every net of a co-trained group takes the step it would take alone from the same state.  "Alone" with the planner
    setting of the co-trained program (`Engine.co_share = K`: every launch gets the grid it has inside the joint launch
    -- resident slots / K), so that the comparison is about the LAUNCH GROUPING and nothing else: the grids fix which
    tiles a workgroup sums before its fp64 atomic (the BatchNorm statistics' fp32 partial sums), the pixel split of the
    weight gradients (the slab count) and the forward body of the deep 4x4 convs (the K-split body, which a lone net
    uses where a launch has too few workgroups, adds two partial sums per output).  Then everything agrees to the last
    bits (hard decisions exactly; sums that meet in fp64 atomics to 1e-5 of the tensor's scale)."""
    import arch_and_hypers as A
    from lib._co import CoTrainer
    ks = A.k_cpts
    mk = {'ac': lambda i: A.ac_chain(k_cpt=ks[i % 8]), 'cr': lambda i: A.cr_chain(k_cpt=ks[i % 8]), 'sr': lambda i: A.sr_chain(8),
          'tree': lambda i: A.ac_tree(k_cpt=ks[(i + 2) % 8])}[kind]        # (the reference's 47-block tree: siblings accumulate into one map)
    co_nets = _nets([mk(i) for i in range(K)])
    solo = _nets([mk(i) for i in range(K)])
    co = CoTrainer(co_nets)
    _compare_with_solo_steps(co_nets, solo, co.run, K, n, routed=kind != 'sr')


def _compare_with_solo_steps(co_nets, solo, run_co, share, n, routed=True, steps=4):
    for b in solo:
        b.engine().co_share = share
    for t in range(steps):                                    # eager, capture + replay, replays
        feeds_co, feeds_so = [], []
        for i, (a, b) in enumerate(zip(co_nets, solo)):
            _copy_state(a, b)                                 # teacher-forced: the solo net starts the step from the co net's state
            x0, y = batch(n, seed=10 * t + i)
            r = routed[i] if isinstance(routed, list) else routed
            extra = {a.τ: 0.5 + 0.1 * i} if r else {}
            extra_b = {b.τ: 0.5 + 0.1 * i} if r else {}
            feeds_co.append({a.x0: x0, a.y: y, a.mode: 'tr', a.λ_lrn: 0.05 / (1 + t), **extra})
            feeds_so.append({b.x0: x0, b.y: y, b.mode: 'tr', b.λ_lrn: 0.05 / (1 + t), **extra_b})
        before = [a.engine().P.clone() for a in co_nets]
        torch.cuda.synchronize()
        run_co(feeds_co)
        for b, f in zip(solo, feeds_so):
            b.train.run(f)
        torch.cuda.synchronize()
        close = lambda u, v: torch.allclose(u, v, rtol=1e-5, atol=1e-9)
        for i, (a, b) in enumerate(zip(co_nets, solo)):
            ea, eb = a.engine(), b.engine()
            for la, lb in zip(a.layers, b.layers):
                assert close(la.p_tr, lb.p_tr) and torch.equal(la.p_ev, lb.p_ev), (t, i, la.name)
            assert close(ea.loss, eb.loss)
            for pa, pb in zip(ea.trainable, eb.trainable):
                ga, gb = pa.grad.double(), pb.grad.double()
                assert float((ga - gb).abs().max()) <= 1e-5 * float(gb.abs().max()) + 1e-12, (t, i, pa.owner.name, pa.name)
            da, db = (ea.P - before[i]).double(), (eb.P - before[i]).double()
            for pa in ea.trainable:
                sl = slice(pa.offset, pa.offset + pa.size)
                ulp = 1.2e-7 * float(ea.P[sl].abs().max())          # (the update is a difference of fp32 parameter values)
                assert float((da[sl] - db[sl]).abs().max()) <= 1e-5 * float(db[sl].abs().max()) + ulp, (t, i, pa.owner.name, pa.name)
            assert torch.allclose(ea.S, eb.S, rtol=1e-6, atol=1e-9), (t, i, 'BatchNorm moving averages')
    # the weight packs the fused optimizer keeps current == a fresh packing of the parameters
    for a in co_nets:
        e = a.engine()
        kept = e.packs.clone()
        e._pack()
        torch.cuda.synchronize()
        assert torch.equal(kept, e.packs)


@pytest.mark.parametrize('makers,streams,sizes,share', [
    ('ac4', 2, [2, 2], 2), ('ac5', 4, [2, 1, 1, 1], 4), ('sr_mixed', 4, [1, 1, 1], 2), ('ac2_tree2_sr', 2, [2, 2, 1], 2), ('ac4', 1, [4], 4)])
def test_groups_side_by_side_equal_the_solo_steps(makers, streams, sizes, share):
    """CoGroups: the nets split into groups of one architecture, each group's joint hipGraph on its own stream, the groups
    free-running side by side (no edges between the graphs).  Nothing of that may show in the results: every net takes
    the step it takes alone with the groups' planner setting (co_share = the groups' share).  Nets of different
    architectures (chains of 2, 5 and 8 blocks, statically routed) run as groups of one."""
    import arch_and_hypers as A
    from lib._co import CoGroups
    ks = A.k_cpts
    mk = {'ac4': lambda: [A.ac_chain(k_cpt=ks[i]) for i in range(4)], 'ac5': lambda: [A.ac_chain(k_cpt=ks[i]) for i in range(5)],
          'sr_mixed': lambda: [A.sr_chain(2), A.sr_chain(5), A.sr_chain(8)],
          'ac2_tree2_sr': lambda: [A.ac_chain(k_cpt=ks[1]), A.ac_chain(k_cpt=ks[2]), A.ac_tree(k_cpt=ks[3]), A.ac_tree(k_cpt=ks[4]), A.sr_chain(3)]}[makers]
    co_nets, solo = _nets(mk()), _nets(mk())
    cg = CoGroups.plan(co_nets, streams=streams)
    assert [c.K for c in cg.groups] == sizes and cg.share == share
    def run(feeds):
        cg.run(feeds)
        cg.join()
    routed = [net._net_kind != 'sr' for net in co_nets]
    _compare_with_solo_steps(co_nets, solo, run, cg.share, 16, routed=routed)


@pytest.mark.calibrate
def test_stream_calibration():
    """concurrent_streams: between one and the requested number of streams whose dependent launch chains overlap (four on
    this runtime's four hardware queues when nothing else loads the GPU -- not asserted: a timing measurement), and a plan
    built on the measurement trains."""
    import arch_and_hypers as A
    from lib._co import CoGroups, concurrent_streams
    found = concurrent_streams(torch.device('cuda:0'), 4)
    assert 1 <= len(found) <= 4 and len({s.cuda_stream for s in found}) == len(found)
    nets = _nets([A.ac_chain(k_cpt=k) for k in A.k_cpts[:4]])
    cg = CoGroups.plan(nets, streams=4)
    assert sum(c.K for c in cg.groups) == 4 and len(cg.streams) <= max(1, len(cg.groups))
    for t in range(3):
        feeds = []
        for i, net in enumerate(nets):
            x0, y = batch(16, seed=i + 10 * t)
            feeds.append({net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: 0.02, net.τ: 0.8})
        cg.run(feeds)
    cg.join()
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(net.engine().P).all()) for net in nets)


def test_groups_side_by_side_are_deterministic():
    """A race detector for the side-by-side form: groups on different streams share nothing but the read-only dataset, and
    every group's joint graph is deterministic -- so 60 free-running rounds (no barrier between steps, the groups drift
    against each other) from one state, twice, must end in bit-identical parameters, momentum and BatchNorm state,
    whatever the interleaving was."""
    import arch_and_hypers as A
    from lib._co import CoGroups
    ks = A.k_cpts
    outs = []
    for rep in range(2):
        nets = _nets([A.ac_chain(k_cpt=ks[i]) for i in range(4)] + [A.sr_chain(3)])
        cg = CoGroups.plan(nets, streams=4)
        assert [c.K for c in cg.groups] == [2, 1, 1, 1]
        xs = [batch(32, seed=50 + i) for i in range(len(nets))]
        for t in range(60):
            feeds = [{net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: 0.02, **({net.τ: 1.0 - 0.01 * t} if net._net_kind != 'sr' else {})}
                     for net, (x0, y) in zip(nets, xs)]
            cg.run(feeds)
        cg.join()
        torch.cuda.synchronize()
        outs.append([(net.engine().P.clone(), net.engine().A.clone(), net.engine().S.clone()) for net in nets])
    for (p0, a0, s0), (p1, a1, s1) in zip(*outs):
        assert torch.equal(p0, p1) and torch.equal(a0, a1) and torch.equal(s0, s1)
        assert bool(torch.isfinite(p0).all())


@pytest.mark.parametrize('kind,K,n', [('ac', 2, 16), ('cr', 3, 16), ('ac', 8, 128)])
def test_cotrained_net_matches_the_oracle(kind, K, n):
    """The whole-net oracle check of tests/test_net_parity.py (decision-forced float64, every gradient and update within
    1e-4 of the tensor's scale, p_ev / delta_cor exact), on one net of a co-trained group -- the others step beside it on
    other batches and other hyper-parameters."""
    import arch_and_hypers as A
    from lib._co import CoTrainer
    mk = A.ac_chain if kind == 'ac' else A.cr_chain
    tau = A.τ_ds if kind == 'ac' else A.τ_cr
    others = _nets([mk(k_cpt=A.k_cpts[(i + 3) % 8]) for i in range(K - 1)], seed0=300)
    state = {'t': 0}

    def stepper(net):
        nets = others[:K // 2] + [net] + others[K // 2:]
        co = CoTrainer(nets)

        def step(feed):
            feeds = []
            for i, o in enumerate(nets):
                if o is net:
                    feeds.append(feed)
                else:
                    x0, y = batch(n, seed=1000 + 10 * state['t'] + i)
                    feeds.append({o.x0: x0, o.y: y, o.mode: 'tr', o.λ_lrn: 0.03, o.τ: tau(2000 * i)})
            state['t'] += 1
            co.run(feeds)
        return step
    run_case(mk(k_cpt=1.6e-8), n, lambda net, t: {net.τ: tau(t * 5000)}, steps=3 if n <= 16 else 2, stepper=stepper)


def test_train_nets_cli_co_train(tmp_path):
    """train-nets --co-train 3: three nets of the experiment advance together through the input pipeline (each net its own
    draws and its own on-device batch assembly as launch 0..2 of the joint graph), log and checkpoint like the serial loop,
    and a statistics pass at 512 images between two steps (the engines' own evaluation programs) does not disturb the
    joint program."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / 'nets')
    cmd = [sys.executable, os.path.join(root, 'multipath-nn_amd', 'train-nets'), 'cifar10-ac', '--synthetic', '--iters', '6',
           '--log-every', '3', '--nets', '0', '2', '5', '--co-train', '3', '--stats-batch', '512', '--out', out]
    subprocess.check_call(cmd, cwd=str(tmp_path))
    base = os.path.join(out, 'cifar10-ac')
    for i in (0, 2, 5):
        for f in ('%.4i.npy', '%.4i-stats.npy', '%.4i-log.txt', '%.4i-stats/00000003.npy', '%.4i-stats/00000006.npy'):
            assert os.path.exists(os.path.join(base, f % i)), f % i
        desc = np.load(os.path.join(base, '%.4i-stats.npy' % i), allow_pickle=True)[()]
        assert desc['type'] == 'ActorNet' and 0 <= desc['stats_ts']['acc'] <= 1


@pytest.mark.parametrize('joint', [True, False])
def test_cotrained_pipeline_feeds_each_net_its_own_batches(joint):
    """Co-trained nets bound to one Dataset: every net trains on the batch staged for IT (its own record buffer), the draws
    come from the one numpy stream in net order, step after step -- also after a larger evaluation batch reallocated one
    engine's input buffers between two steps (the joint program is rebuilt).  joint: ONE gather launch for the whole group
    (Dataset.bind_cotrainer, mpnn_augment_batch_multi) instead of one per net."""
    import arch_and_hypers as A
    from lib._co import CoTrainer
    from lib.data import Dataset
    ds = Dataset.synthetic(n_tr=300, n_ts=140, seed=1)
    ref = Dataset.synthetic(n_tr=300, n_ts=140, seed=1)
    nets = _nets([A.ac_chain(k_cpt=k) for k in (0.0, 1e-9, 4e-9)])
    engs = [net.engine() for net in nets]
    n = 32
    co = CoTrainer(nets)
    bound = ds.bind_cotrainer(co, n) if joint else [ds.bind_engine(e, n) for e in engs]
    np.random.seed(11)
    want = []
    state0 = np.random.get_state()
    for t in range(5):
        for k in range(3):
            want.append(ref.augmented_training_batch(n))
    np.random.set_state(state0)
    for t in range(5):
        feeds = []
        if joint:
            ds.stage_cotrainer_draws(co)
        for net, e, (x0, y) in zip(nets, engs, bound):
            if not joint:
                ds.stage_training_draws(n, eng=e)
            feeds.append({net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: 0.01, net.τ: 1.0})
        co.run(feeds)
        torch.cuda.synchronize()
        for k, e in enumerate(engs):
            wx, wy = want[3 * t + k]
            assert np.abs(e.x0[:n].cpu().numpy() - wx).max() <= 1e-6, (t, k)
            assert np.array_equal(e.y[:n].cpu().numpy(), wy), (t, k)
        if t == 2:
            xb = torch.rand(n * 8, *ds.x0_shape, device='cuda')
            yb = torch.zeros(n * 8, ds.y_shape[0], device='cuda'); yb[:, 0] = 1
            nets[1].eval({nets[1].x0: xb, nets[1].y: yb})


@pytest.mark.parametrize('draws', ['joint', 'serial'])
def test_groups_pipeline_feeds_each_net_its_own_batches(draws):
    """CoGroups through the input pipeline, as train-nets --co-train runs it: one gather launch and one record upload per
    GROUP, queued on the group's stream.  joint: the draws come from the one numpy stream in net order, iteration after
    iteration.  serial: every net has its own copy of the stream (lib.data.DrawStream), advanced over all iterations of
    the nets in front of it -- it trains on exactly the batches the SERIAL experiment loop (net after net) feeds it."""
    import arch_and_hypers as A
    from lib._co import CoGroups
    from lib.data import Dataset
    ds = Dataset.synthetic(n_tr=300, n_ts=140, seed=1)
    ref = Dataset.synthetic(n_tr=300, n_ts=140, seed=1)
    ds.m_sym = ref.m_sym = np.array([1, 0, 1, 1, 0, 0, 1, 0, 1, 1], bool)       # (some classes without the flip draw)
    nets = _nets([A.ac_chain(k_cpt=k) for k in (0.0, 1e-9, 4e-9, 8e-9)] + [A.sr_chain(2)])
    engs = [net.engine() for net in nets]
    n, K, T = 32, len(nets), 4
    ds.to_device('cuda:0')
    cg = CoGroups.plan(nets, streams=2)
    assert [c.K for c in cg.groups] == [2, 2, 1]
    bound = [None] * K

    def bind(g, co, span):
        bound[span[0]:span[1]] = ds.bind_cotrainer(co, n)
    cg.on_group_streams(bind)
    np.random.seed(11)
    state0 = np.random.get_state()
    if draws == 'joint':
        seq = [ref.augmented_training_batch(n) for _ in range(T * K)]
        want = lambda t, k: seq[K * t + k]
        streams = None
    else:
        seq = [ref.augmented_training_batch(n) for _ in range(K * T)]           # the serial loop: net 0's T batches, then net 1's ...
        want = lambda t, k: seq[T * k + t]
        streams = ds.serial_positions(list(range(K)), T, n, seed=11)
    np.random.set_state(state0)
    for t in range(T):
        feeds = [{net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: 0.01, **({net.τ: 1.0} if net._net_kind != 'sr' else {})}
                 for net, (x0, y) in zip(nets, bound)]

        def step(g, co, span):
            ds.stage_cotrainer_draws(co, streams=None if streams is None else streams[span[0]:span[1]])
            co.run(feeds[span[0]:span[1]])
        cg.on_group_streams(step)
        cg.join()
        torch.cuda.synchronize()
        for k, e in enumerate(engs):
            wx, wy = want(t, k)
            assert np.abs(e.x0[:n].cpu().numpy() - wx).max() <= 1e-6, (t, k)
            assert np.array_equal(e.y[:n].cpu().numpy(), wy), (t, k)


def test_train_nets_cli_co_train_mixed_architectures(tmp_path):
    """train-nets cifar10-sr --co-train 3: chains of 1, 2 and 4 blocks cannot share launches; they run side by side, each
    net's own hipGraph on its own stream, and log / checkpoint like the serial loop."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / 'nets')
    cmd = [sys.executable, os.path.join(root, 'multipath-nn_amd', 'train-nets'), 'cifar10-sr', '--synthetic', '--iters', '4',
           '--log-every', '2', '--nets', '0', '1', '3', '--co-train', '3', '--stats-batch', '256', '--out', out]
    subprocess.check_call(cmd, cwd=str(tmp_path))
    base = os.path.join(out, 'cifar10-sr')
    for i in (0, 1, 3):
        for f in ('%.4i.npy', '%.4i-stats.npy', '%.4i-log.txt', '%.4i-stats/00000002.npy', '%.4i-stats/00000004.npy'):
            assert os.path.exists(os.path.join(base, f % i)), f % i
        desc = np.load(os.path.join(base, '%.4i-stats.npy' % i), allow_pickle=True)[()]
        assert desc['type'] == 'SRNet' and 0 <= desc['stats_ts']['acc'] <= 1


def test_train_nets_shard_nets_under_torchrun(tmp_path):
    """train-nets --shard-nets under torchrun with two ranks (both on GPU 0 here): the nets of the experiment are dealt to
    the ranks -- rank 0 trains nets 0 and 2 (co-trained), rank 1 net 1 -- with no process group and no collective; every
    rank writes the files of its own nets."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / 'nets')
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MPNN_DP_ONE_GPU='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'multipath-nn_amd', 'train-nets'), 'cifar10-cr', '--synthetic',
           '--iters', '4', '--log-every', '4', '--nets', '0', '1', '2', '--shard-nets', '--co-train', '2', '--stats-batch', '256',
           '--out', out]
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    base = os.path.join(out, 'cifar10-cr')
    for i in (0, 1, 2):
        for f in ('%.4i.npy', '%.4i-stats.npy', '%.4i-log.txt'):
            assert os.path.exists(os.path.join(base, f % i)), f % i


@pytest.mark.parametrize('seed', [int(s) for s in __import__('os').environ.get('MPNN_FUZZ_COTREE_SEEDS', '0 1').split()])
def test_cotrained_random_trees_and_architectures_equal_the_solo_steps(seed):
    """Co-training beyond the shipped shapes: three nets of one RANDOMLY drawn tree (forks, static links:
    tests/test_fuzz_trees_gpu.py) or of one randomly drawn channel / scale table (tests/test_fuzz_arch_gpu.py), in one
    joint graph and as groups on streams -- every net takes the step it takes alone with the group's planner setting."""
    from lib._co import CoGroups, CoTrainer
    from lib.net_types import ActorNet, CriticNet
    from test_fuzz_arch_gpu import draw_arch, make_chain
    from test_fuzz_trees_gpu import draw_tree, make_tree
    rng = np.random.default_rng(1700 + seed)
    kind = (ActorNet, CriticNet)[seed % 2]
    if seed % 3 == 0:
        arch = draw_arch(rng)
        mk = lambda k: make_chain(kind, arch, len(arch[0]), k_cpt=k)
    else:
        has_switch = lambda t: (int(t[0]) + len(t[1]) >= 2) or any(has_switch(c) for c in t[1])
        spec = draw_tree(rng, int(rng.integers(2, 5)))
        while not has_switch(spec):
            spec = draw_tree(rng, int(rng.integers(2, 5)))
        mk = lambda k: make_tree(kind, spec, k_cpt=k)
    ks = [0.0, 4e-9, 1.6e-8]
    n = int(rng.choice([8, 24, 128]))
    co_nets, solo = _nets([mk(k) for k in ks]), _nets([mk(k) for k in ks])
    if co_nets[0].engine()._groupable():
        co = CoTrainer(co_nets)
        _compare_with_solo_steps(co_nets, solo, co.run, 3, n, steps=3)
        co_nets, solo = _nets([mk(k) for k in ks]), _nets([mk(k) for k in ks])
    cg = CoGroups.plan(co_nets, streams=2)          # (an architecture without multi-net launch forms: groups of one)
    assert co_nets[0].engine()._groupable() or [c.K for c in cg.groups] == [1, 1, 1]

    def run(feeds):
        cg.run(feeds)
        cg.join()
    _compare_with_solo_steps(co_nets, solo, run, cg.share, n, steps=3)


@pytest.mark.parametrize('kind,K,S', [('ac', 3, 4), ('cr', 2, 3), ('ac', 1, 2)])
def test_k_step_joint_replay_equals_k_joint_steps(kind, K, S):
    """CoTrainer.run_steps: S joint steps captured as ONE hipGraph (per-step schedule values from a device ring, copied into
    every net's hyp row by the step's own mpnn_exit_tail_fwd) == S calls of run(), BIT FOR BIT -- parameters, momentum,
    BatchNorm state of every net -- on the engines' resident input buffers (different learning rates / temperatures per
    step and per net, so a wrong or stale row would show)."""
    import arch_and_hypers as A
    from lib._co import CoTrainer
    ks = A.k_cpts
    mk = {'ac': lambda i: A.ac_chain(k_cpt=ks[i % 8]), 'cr': lambda i: A.cr_chain(k_cpt=ks[i % 8])}[kind]
    a_nets, b_nets = _nets([mk(i) for i in range(K)]), _nets([mk(i) for i in range(K)])
    n = 32
    for nets in (a_nets, b_nets):
        for i, net in enumerate(nets):
            e = net.engine()
            e._ensure_capacity(n)
            x0, y = batch(n, seed=40 + i)
            e.x0[:n].copy_(torch.from_numpy(x0)); e.y[:n].copy_(torch.from_numpy(y))
    co_a, co_b = CoTrainer(a_nets), CoTrainer(b_nets)

    def feeds(nets, t):
        return [{net.x0: net.engine().x0[:n], net.y: net.engine().y[:n], net.mode: 'tr', net.λ_lrn: 0.05 / (1 + t) * (1 + 0.1 * i),
                 net.τ: 0.5 + 0.07 * t + 0.1 * i} for i, net in enumerate(nets)]
    for rnd in range(4):                                      # warm-up, capture + replay, replays
        co_a.run_steps([feeds(a_nets, rnd * S + j) for j in range(S)])
        for j in range(S):
            co_b.run(feeds(b_nets, rnd * S + j))
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(a_nets, b_nets)):
            ea, eb = a.engine(), b.engine()
            assert torch.equal(ea.P, eb.P) and torch.equal(ea.A, eb.A) and torch.equal(ea.S, eb.S), (rnd, i)
            for la, lb in zip(a.layers, b.layers):
                assert torch.equal(la.p_tr, lb.p_tr) and torch.equal(la.p_ev, lb.p_ev), (rnd, i, la.name)
    assert any(k[0] == 'K' and not isinstance(v, str) for k, v in co_a._graphs.items() if isinstance(k, tuple)), 'no K-step graph was captured'


def test_k_step_joint_graph_is_dropped_when_an_engine_reallocates():
    """An evaluation at a larger batch between two K-step joint replays reallocates an engine's buffers: the merged program
    and every joint graph are rebuilt (CoTrainer._program compares the engines' generations BEFORE run_steps looks its
    graph up) -- the next replay must not run the old graph over the old buffers.  Same results as single joint steps."""
    import arch_and_hypers as A
    from lib._co import CoTrainer
    K, S, n = 2, 3, 16
    mk = lambda i: A.ac_chain(k_cpt=A.k_cpts[i % 8])
    a_nets, b_nets = _nets([mk(i) for i in range(K)]), _nets([mk(i) for i in range(K)])

    def fill(nets):
        for i, net in enumerate(nets):
            e = net.engine()
            e.ensure_capacity(n)
            x0, y = batch(n, seed=60 + i)
            e.x0[:n].copy_(torch.from_numpy(x0)); e.y[:n].copy_(torch.from_numpy(y))
    fill(a_nets); fill(b_nets)
    co_a, co_b = CoTrainer(a_nets), CoTrainer(b_nets)

    def feeds(nets, t):
        return [{net.x0: net.engine().x0[:n], net.y: net.engine().y[:n], net.mode: 'tr', net.λ_lrn: 0.05 / (1 + t),
                 net.τ: 0.6 + 0.05 * t + 0.1 * i} for i, net in enumerate(nets)]

    def rounds(r0, r1):
        for rnd in range(r0, r1):
            co_a.run_steps([feeds(a_nets, rnd * S + j) for j in range(S)])
            for j in range(S):
                co_b.run(feeds(b_nets, rnd * S + j))
            torch.cuda.synchronize()
            for i, (a, b) in enumerate(zip(a_nets, b_nets)):
                ea, eb = a.engine(), b.engine()
                assert torch.equal(ea.P, eb.P) and torch.equal(ea.A, eb.A) and torch.equal(ea.S, eb.S), (rnd, i)
    rounds(0, 3)                                              # warm-up, capture + replay, a replay
    gens = [net.engine().generation for net in a_nets]
    for nets in (a_nets, b_nets):                             # a larger evaluation batch: net 0's buffers move
        net = nets[0]
        x0, y = batch(300, seed=7)
        net.eval({net.x0: x0, net.y: y, net.τ: 0.5})
    assert a_nets[0].engine().generation != gens[0], 'the evaluation did not reallocate'
    fill(a_nets); fill(b_nets)                                # (the resident input buffers are new ones)
    rounds(3, 6)


def test_k_step_joint_replay_through_the_input_pipeline():
    """... and with the input pipeline bound (Dataset.bind_cotrainer): launch 0 of step j of the joint graph gathers every
    net's batch from record slot j; stage_cotrainer_draws_k draws the S x K batches (every net from its own DrawStream: the
    batches of the serial loop) and uploads them at once.  Same parameters, bit for bit, as single joint steps fed from the
    same streams."""
    import arch_and_hypers as A
    from lib._co import CoTrainer
    from lib.data import Dataset
    K, S, n, T = 3, 4, 32, 12
    runs = []
    for form in ('k', 'single'):
        ds = Dataset.synthetic(n_tr=300, n_ts=140, seed=1)
        ds.to_device('cuda:0')
        nets = _nets([A.ac_chain(k_cpt=k) for k in (0.0, 1e-9, 4e-9)])
        co = CoTrainer(nets)
        bound = ds.bind_cotrainer(co, n)
        streams = ds.serial_positions(list(range(K)), T, n, seed=11)
        feeds = lambda t: [{net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: 0.05 / (1 + t), net.τ: 1.0 - 0.02 * t}
                           for net, (x0, y) in zip(nets, bound)]
        t = 0
        while t < T:
            if form == 'k':
                ds.stage_cotrainer_draws_k(co, S, streams=streams)
                co.run_steps([feeds(t + j) for j in range(S)])
                t += S
            else:
                ds.stage_cotrainer_draws(co, streams=streams)
                co.run(feeds(t))
                t += 1
        torch.cuda.synchronize()
        runs.append([(net.engine().P.clone(), net.engine().S.clone(), net.engine().x0[:n].clone()) for net in nets])
    for i, (a, b) in enumerate(zip(*runs)):
        assert torch.equal(a[2], b[2]), ('the last batch of net %d differs' % i)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), i
