"""GPU: a long-lived engine against fresh ones -- stale-state fuzz.

An engine keeps a lot between calls: buffers sized for the largest batch it has seen, programs and hipGraphs per (mode,
batch), accumulators that one step leaves cleared for the next, weight packs the optimizer keeps current, sample lists
of the routed evaluation, result views.  Every bug of that kind found so far (input views orphaned by a reallocation,
a weight-gradient split sized for the capacity instead of the batch, a prologue missing from one graph form) was
invisible to tests that build a fresh engine per case.  Here ONE engine runs a random sequence of operations --
training steps at several batch sizes (eager, captured, replayed, K steps per graph), dense and routed evaluation at
sizes on both sides of its capacity, a mode-'tr' forward without the train op -- and after every operation a FRESH
net, loaded with the state the long-lived one had before the operation, performs the same operation: the kernels are
deterministic and the planner depends on the batch only, so the two must agree BIT FOR BIT (parameters, momentum,
BatchNorm state, losses; p_ev / c_err / delta_cor / router outputs)."""
import numpy as np
import pytest
import torch

from test_net_parity import batch, perturb_routers

pytestmark = pytest.mark.gpu


def _make(kind):
    import arch_and_hypers as A
    mk = {'ac': lambda: A.ac_chain(k_cpt=1.6e-8), 'cr': lambda: A.cr_chain(k_cpt=4e-9, optimistic=True),
          'sr': lambda: A.sr_chain(5), 'tree': lambda: A.ac_tree(k_cpt=1e-9), 'dyn': lambda: A.cr_chain(dyn_k_cpt=True)}[kind]()
    net = mk((32, 32, 3), (10,))
    net.engine().init_params(77)
    if net._net_kind != 'sr':
        perturb_routers(net, seed=3)
    return net


def _state(net):
    e = net.engine()
    return e.P.clone(), e.A.clone(), e.S.clone()


def _load(net, st):
    e = net.engine()
    for dst, src in zip((e.P, e.A, e.S), st):
        dst.copy_(src)
    e.invalidate_packs()


def _results(net, train):
    e = net.engine()
    torch.cuda.synchronize()
    out = {'P': e.P.clone(), 'A': e.A.clone(), 'S': e.S.clone()}
    if train:
        out['loss'] = e.loss.clone()
    for nd in e.nodes:                                # (what the mode writes: the rest of a long-lived engine's buffers is history)
        if train and net._net_kind != 'sr':
            out[('p_tr', nd.idx)] = nd.layer.p_tr.clone()
        if not train:
            out[('p_ev', nd.idx)] = nd.layer.p_ev.clone()
    for nd in e.leaves:
        out[('c_err', nd.idx)] = nd.layer.c_err.clone()
        if not train:
            out[('d_cor', nd.idx)] = nd.layer.δ_cor.clone()
    if not train:
        for nd in e.switches:
            out[('r', nd.idx)] = nd.layer.router.x.clone()
    return out


def _apply(net, op, t):
    what, n, arg = op
    x0, y = batch(n, seed=1000 + 17 * t)
    τ = {} if net._net_kind == 'sr' else {net.τ: 0.7 + 0.05 * (t % 5)}
    if getattr(net.hypers, 'dyn_k_cpt', False):           # per-sample k_cpt: a vector with every feed
        import arch_and_hypers as A
        kvec = lambda j: np.random.default_rng(31 * t + j).choice(A.k_cpts, n).astype(np.float32)
    else:
        kvec = None
    if kvec is not None and what != 'steps':
        τ[net.k_cpt] = kvec(0)
    if what == 'train':
        for rep in range(arg):                        # (1: whatever comes next of eager / capture / replay; 3: all of them)
            net.train.run({net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: 0.03 / (1 + rep), **τ})
        return True
    if what == 'steps':
        e = net.engine()
        e._ensure_capacity(n)
        e.x0[:n].copy_(torch.from_numpy(x0)); e.y[:n].copy_(torch.from_numpy(y))
        for rep in range(3):                          # warm, capture, replay of the K-step graph
            net.train.run_steps([{net.x0: e.x0[:n], net.y: e.y[:n], net.mode: 'tr', net.λ_lrn: 0.02 / (1 + j), **τ,
                                  **({net.k_cpt: kvec(j)} if kvec is not None else {})} for j in range(arg)])
        return True
    if what == 'eval':
        net.eval({net.x0: x0, net.y: y, **({net.k_cpt: kvec(0)} if kvec is not None else {})}, routed=arg)
        return False
    if what == 'fwd_tr':                              # a fetch in mode 'tr' without the train op (moves the moving averages)
        net.eval({net.x0: x0, net.y: y, net.mode: 'tr', **τ})
        return False
    raise ValueError(what)


def _draw_ops(rng, kind, count):
    ops = []
    for _ in range(count):
        u = rng.random()
        if u < 0.4:
            ops.append(('train', int(rng.choice([3, 16, 37, 128, 128, 200] if kind != 'tree' else [3, 16, 37, 128])), int(rng.choice([1, 1, 3]))))
        elif u < 0.5 and kind != 'tree':
            ops.append(('steps', int(rng.choice([16, 128])), int(rng.choice([2, 3]))))
        elif u < 0.9:
            n = int(rng.choice([5, 64, 200, 600, 1500] if kind != 'tree' else [5, 64, 200, 600]))
            routed = False if kind == 'sr' else [False, True, 1, 2, 4, 'auto'][int(rng.integers(0, 6))]
            ops.append(('eval', n, routed))
        else:
            ops.append(('fwd_tr', int(rng.choice([16, 40])), None))
    return ops


import os
_CASES = [('ac', 0), ('ac', 1), ('cr', 2), ('sr', 3), ('tree', 4), ('cr', 5), ('ac', 6), ('tree', 7), ('dyn', 8), ('dyn', 9)]
if os.environ.get('MPNN_STATE_FUZZ_SEEDS'):           # (a longer hunt: MPNN_STATE_FUZZ_SEEDS="100 140" -> seeds 100 .. 139)
    lo, hi = (int(v) for v in os.environ['MPNN_STATE_FUZZ_SEEDS'].split())
    _CASES = [(('ac', 'cr', 'sr', 'tree', 'dyn')[s % 5], s) for s in range(lo, hi)]


@pytest.mark.parametrize('kind,seed', _CASES)
def test_long_lived_engine_equals_fresh_engines(kind, seed):
    rng = np.random.default_rng(seed)
    ops = _draw_ops(rng, kind, 9 if kind != 'tree' else 6)
    long_lived = _make(kind)
    for t, op in enumerate(ops):
        before = _state(long_lived)
        train = _apply(long_lived, op, t)
        got = _results(long_lived, train)
        fresh = _make(kind)
        _load(fresh, before)
        _apply(fresh, op, t)
        want = _results(fresh, train)
        for key in want:
            assert torch.equal(got[key], want[key]), (kind, seed, t, op, key, float((got[key].double() - want[key].double()).abs().max()))
        del fresh


# ---------------------------------------------------------------------------------------------------------------------
_MORE = int(os.environ.get('MPNN_STATE_FUZZ_MORE', '0'))


@pytest.mark.parametrize('seed', list(range(_MORE)) or [0, 1, 2])
def test_long_lived_conv_engine_equals_fresh_engines(seed):
    """The same for the single-scale Conv engine (lib/_plan_conv.py: capacity 128 at construction, reallocation beyond):
    training steps at 3 ... 130 samples and evaluations at 7 ... 200 in a random order."""
    from test_conv_layer import conv_net, pooled_conv_net
    rng = np.random.default_rng(40 + seed)
    mk = (conv_net(res=True), pooled_conv_net(), conv_net(res=False))[seed % 3]

    def make():
        net = mk((16, 16, 3), (10,))
        net.engine().init_params(9)
        return net

    def apply(net, op, t):
        what, n = op
        g = np.random.default_rng(2000 + t)
        x0 = g.random((n, 16, 16, 3)).astype(np.float32)
        y = np.eye(10, dtype=np.float32)[g.integers(0, 10, n)]
        if what == 'train':
            net.train.run({net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: 0.05, net.μ_lrn: 0.9})
        else:
            net.eval({net.x0: x0, net.y: y})

    def results(net):
        e = net.engine()
        torch.cuda.synchronize()
        out = {'P': e.P.clone(), 'A': e.A.clone()}
        out.update({('state', k[1]): v.clone() for k, v in net.state().items()})
        return out
    ops = [(('train', int(rng.choice([3, 5, 24, 64, 130]))) if rng.random() < 0.6 else ('eval', int(rng.choice([7, 100, 200])))) for _ in range(8)]
    long_lived = make()
    for t, op in enumerate(ops):
        e = long_lived.engine()
        before = (e.P.clone(), e.A.clone())
        apply(long_lived, op, t)
        got = results(long_lived)
        fresh = make()
        fresh.engine().P.copy_(before[0]); fresh.engine().A.copy_(before[1])
        apply(fresh, op, t)
        want = results(fresh)
        for key in want:
            # 1x1 convolutions add their weight gradients with fp32 atomics (csrc/conv_nhwc.hip): the order of the addends varies
            tol = 0.0 if op[0] == 'eval' else 2e-6
            d = float((got[key].double() - want[key].double()).abs().max())
            assert d <= tol * (1.0 + float(want[key].double().abs().max())), (seed, t, op, key, d)


@pytest.mark.parametrize('seed', list(range(_MORE)) or [0, 1])
def test_long_lived_cotrainer_equals_fresh_ones(seed):
    """A co-trained group that lives through joint steps, SOLO steps of one member (its row of the joint schedule buffer is
    rewritten), evaluations that reallocate a member's buffers (the joint program is rebuilt) and K-step graphs of a
    member, against fresh groups loaded with the state before each operation: bit for bit."""
    import arch_and_hypers as A
    from lib._co import CoTrainer
    rng = np.random.default_rng(70 + seed)
    K, n = 3, (16, 16, 128, 37)[seed % 4]

    def make():
        nets = []
        for i in range(K):
            net = (A.ac_chain if seed % 2 == 0 else A.cr_chain)(k_cpt=A.k_cpts[i + 1])((32, 32, 3), (10,))
            net.engine().init_params(30 + i)
            perturb_routers(net, seed=8 + i)
            nets.append(net)
        return nets, CoTrainer(nets)

    def feed(net, t, i):
        x0, y = batch(n, seed=3000 + 10 * t + i)
        return {net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: 0.04 / (1 + t % 3), net.τ: 0.6 + 0.1 * i}

    def apply(nets, co, op, t):
        what, arg = op
        if what == 'joint':
            for rep in range(arg):
                co.run([feed(net, t + rep, i) for i, net in enumerate(nets)])
        elif what == 'solo':
            nets[arg].train.run(feed(nets[arg], t, arg))
        elif what == 'eval':
            x0, y = batch(arg[1], seed=4000 + t)
            nets[arg[0]].eval({nets[arg[0]].x0: x0, nets[arg[0]].y: y}, routed=bool(t % 2))
        elif what == 'steps':
            net = nets[arg]; e = net.engine()
            x0, y = batch(n, seed=5000 + t)
            e._ensure_capacity(n)
            e.x0[:n].copy_(torch.from_numpy(x0)); e.y[:n].copy_(torch.from_numpy(y))
            for rep in range(3):
                net.train.run_steps([{net.x0: e.x0[:n], net.y: e.y[:n], net.mode: 'tr', net.λ_lrn: 0.02, net.τ: 0.9} for _ in range(2)])

    def results(nets):
        torch.cuda.synchronize()
        out = {}
        for i, net in enumerate(nets):
            e = net.engine()
            out.update({('P', i): e.P.clone(), ('A', i): e.A.clone(), ('S', i): e.S.clone()})
        return out
    ops = []
    for _ in range(9):
        u = rng.random()
        ops.append(('joint', int(rng.choice([1, 3]))) if u < 0.5 else ('solo', int(rng.integers(0, K))) if u < 0.65 else
                   ('eval', (int(rng.integers(0, K)), int(rng.choice([8, 300])))) if u < 0.85 else ('steps', int(rng.integers(0, K))))
    ops.append(('joint', 2))
    nets, co = make()
    for t, op in enumerate(ops):
        before = [_state(net) for net in nets]
        apply(nets, co, op, t)
        got = results(nets)
        fnets, fco = make()
        for net, st in zip(fnets, before):
            _load(net, st)
        apply(fnets, fco, op, t)
        want = results(fnets)
        for key in want:
            assert torch.equal(got[key], want[key]), (seed, t, op, key, float((got[key].double() - want[key].double()).abs().max()))


@pytest.mark.parametrize('seed', list(range(_MORE)) or [0, 1, 2])
def test_bound_input_pipeline_through_a_random_sequence(seed):
    """Dataset.bind_engine under a random mix of single steps, K-step graphs (record slots 0 .. K-1, one upload), private
    draw streams (DrawStream) and evaluations that reallocate the engine's input buffers: after every call the engine's
    input buffers hold exactly the batch the reference's draw sequence assigns to the LAST step of that call."""
    import arch_and_hypers as A
    from lib.data import Dataset, DrawStream
    rng = np.random.default_rng(90 + seed)
    n = int(rng.choice([16, 32]))
    ds = Dataset.synthetic(n_tr=300, n_ts=140, seed=1)
    ref = Dataset.synthetic(n_tr=300, n_ts=140, seed=1)
    ds.m_sym = ref.m_sym = np.array([1, 0, 1, 1, 0, 0, 1, 0, 1, 1], bool)
    net = (A.ac_chain(k_cpt=1e-9), A.sr_chain(3), A.cr_chain(k_cpt=4e-9))[seed % 3]((32, 32, 3), (10,))
    eng = net.engine()
    eng.init_params(3)
    if net._net_kind != 'sr':
        perturb_routers(net, seed=2)
    ds.to_device('cuda:0')
    x0, y = ds.bind_engine(eng, n)
    private = bool(seed % 2)                             # the engine's batches from a private copy of the stream
    stream = DrawStream(123 + seed) if private else None
    ops, total = [], 0
    for _ in range(10):
        u = rng.random()
        if u < 0.45:
            ops.append(('one', 1)); total += 1
        elif u < 0.8:
            k = int(rng.integers(2, 5)); ops.append(('k', k)); total += k
        else:
            ops.append(('eval', int(rng.choice([8, 200, 400]))))
    np.random.seed(123 + seed)
    want = [ref.augmented_training_batch(n) for _ in range(total)]
    np.random.seed(123 + seed)
    τ = {} if net._net_kind == 'sr' else {net.τ: 0.8}
    done = 0
    for t, (what, arg) in enumerate(ops):
        if what == 'eval':
            xb, yb = batch(arg, seed=t)
            net.eval({net.x0: xb, net.y: yb})
            continue
        if what == 'one' and not private:
            ds.stage_training_draws(n, eng=eng)
            net.train.run({net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: 0.01, **τ})
        else:
            ds.stage_training_draws_k(arg, n, eng=eng, stream=stream)
            net.train.run_steps([{net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: 0.01, **τ} for _ in range(arg)])
        done += arg
        torch.cuda.synchronize()
        wx, wy = want[done - 1]
        assert np.abs(eng.x0[:n].cpu().numpy() - wx).max() <= 1e-6, (seed, t, what, arg)
        assert np.array_equal(eng.y[:n].cpu().numpy(), wy), (seed, t, what, arg)
    assert bool(torch.isfinite(eng.P).all())


@pytest.mark.parametrize('seed', list(range(_MORE)) or [0, 1])
def test_long_lived_groups_on_streams_equal_fresh_ones(seed):
    """CoGroups (groups side by side on streams) through free-running rounds, joins, evaluations on the caller's stream
    that reallocate a member's buffers, solo steps of a member and more rounds -- against fresh groups loaded with the
    state before each operation.  A missing fork / join, or a buffer freed under a running group, shows as a mismatch."""
    import arch_and_hypers as A
    from lib._co import CoGroups
    rng = np.random.default_rng(110 + seed)
    n = (16, 32)[seed % 2]

    def make():
        nets = [A.ac_chain(k_cpt=A.k_cpts[i])((32, 32, 3), (10,)) for i in range(4)] + [A.sr_chain(2 + seed % 3)((32, 32, 3), (10,))]
        for i, net in enumerate(nets):
            net.engine().init_params(50 + i)
            if net._net_kind != 'sr':
                perturb_routers(net, seed=4 + i)
        return nets, CoGroups.plan(nets, streams=(2, 4)[seed % 2])

    def feeds(nets, t):
        out = []
        for i, net in enumerate(nets):
            x0, y = batch(n, seed=6000 + 10 * t + i)
            out.append({net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: 0.03, **({net.τ: 0.7} if net._net_kind != 'sr' else {})})
        return out

    def apply(nets, cg, op, t):
        what, arg = op
        if what == 'rounds':
            for r in range(arg):
                cg.run(feeds(nets, t + r))
            cg.join()
        elif what == 'eval':
            x0, y = batch(arg[1], seed=7000 + t)
            nets[arg[0]].eval({nets[arg[0]].x0: x0, nets[arg[0]].y: y})
        elif what == 'solo':
            f = feeds(nets, t)[arg]
            nets[arg].train.run(f)

    ops = [('rounds', 2)]
    for _ in range(6):
        u = rng.random()
        ops.append(('rounds', int(rng.choice([1, 3]))) if u < 0.5 else ('eval', (int(rng.integers(0, 5)), int(rng.choice([8, 200])))) if u < 0.8
                   else ('solo', int(rng.integers(0, 5))))
    ops.append(('rounds', 2))
    nets, cg = make()
    for t, op in enumerate(ops):
        torch.cuda.synchronize()
        before = [_state(net) for net in nets]
        apply(nets, cg, op, t)
        torch.cuda.synchronize()
        got = [_state(net) for net in nets]
        fnets, fcg = make()
        for net, st in zip(fnets, before):
            _load(net, st)
        apply(fnets, fcg, op, t)
        torch.cuda.synchronize()
        want = [_state(net) for net in fnets]
        for i, (g, w) in enumerate(zip(got, want)):
            for name, a, b in zip('PAS', g, w):
                assert torch.equal(a, b), (seed, t, op, i, name, float((a.double() - b.double()).abs().max()))
