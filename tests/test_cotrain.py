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


@pytest.mark.parametrize('kind,K,n', [('ac', 3, 32), ('cr', 2, 16), ('ac', 8, 128), ('sr', 3, 16)])
def test_cotrained_step_equals_the_solo_step(kind, K, n):
    """Every net of a co-trained group takes the step it would take alone from the same state: routing, costs and exit
    gradients bit for bit (the same kernels on the same records), conv weight gradients to fp32 summation order (the
    planner gives each net's weight-gradient launch 1/K of the slots, so the pixel split -- the slab count -- differs)."""
    import arch_and_hypers as A
    from lib._co import CoTrainer
    ks = A.k_cpts
    mk = {'ac': lambda i: A.ac_chain(k_cpt=ks[i % 8]), 'cr': lambda i: A.cr_chain(k_cpt=ks[i % 8]), 'sr': lambda i: A.sr_chain(8)}[kind]
    co_nets = _nets([mk(i) for i in range(K)])
    solo = _nets([mk(i) for i in range(K)])
    co = CoTrainer(co_nets)
    for t in range(4):                                        # eager, capture + replay, replays
        feeds_co, feeds_so = [], []
        for i, (a, b) in enumerate(zip(co_nets, solo)):
            _copy_state(a, b)                                 # teacher-forced: the solo net starts the step from the co net's state
            x0, y = batch(n, seed=10 * t + i)
            extra = {} if kind == 'sr' else {a.τ: 0.5 + 0.1 * i}
            extra_b = {} if kind == 'sr' else {b.τ: 0.5 + 0.1 * i}
            feeds_co.append({a.x0: x0, a.y: y, a.mode: 'tr', a.λ_lrn: 0.05 / (1 + t), **extra})
            feeds_so.append({b.x0: x0, b.y: y, b.mode: 'tr', b.λ_lrn: 0.05 / (1 + t), **extra_b})
        before = [a.engine().P.clone() for a in co_nets]
        co.run(feeds_co)
        for b, f in zip(solo, feeds_so):
            b.train.run(f)
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(co_nets, solo)):
            ea, eb = a.engine(), b.engine()
            for la, lb in zip(a.layers, b.layers):
                assert torch.equal(la.p_tr, lb.p_tr) and torch.equal(la.p_ev, lb.p_ev), (t, i, la.name)
            assert torch.equal(ea.loss, eb.loss) or kind == 'sr'
            for pa, pb in zip(ea.trainable, eb.trainable):
                ga, gb = pa.grad.cpu().numpy().astype(np.float64), pb.grad.cpu().numpy().astype(np.float64)
                assert np.abs(ga - gb).max() <= 2e-5 * np.abs(gb).max() + 1e-7, (t, i, pa.owner.name, pa.name)
            da = (ea.P - before[i]).cpu().numpy().astype(np.float64)
            db = (eb.P - before[i]).cpu().numpy().astype(np.float64)
            for pa in ea.trainable:
                sl = slice(pa.offset, pa.offset + pa.size)
                assert np.abs(da[sl] - db[sl]).max() <= 1e-4 * np.abs(db[sl]).max() + 1e-7, (t, i, pa.owner.name, pa.name)
            assert torch.allclose(ea.S, eb.S, rtol=1e-5, atol=1e-7), (t, i, 'BatchNorm moving averages')
            assert torch.equal(ea.A[:ea.stat_pad], eb.A[:eb.stat_pad])
    # the weight packs the fused optimizer keeps current == a fresh packing of the parameters
    for a in co_nets:
        e = a.engine()
        kept = e.packs.clone()
        e._pack()
        torch.cuda.synchronize()
        assert torch.equal(kept, e.packs)


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
