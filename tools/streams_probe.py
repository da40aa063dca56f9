#!/usr/bin/env python3
"""K independent nets, each with its OWN one-step hipGraph on its OWN stream, replayed side by side -- no cross-stream
edges, so none of the 20-30 us a graph edge costs.  Against the serial loop and against lib/_co.py's joint
launches.  Works for nets of DIFFERENT architectures (the *-sr experiments: chains of 1 ... 8 blocks), which co-training
refuses.     python tools/streams_probe.py [ac|sr] [share ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'multipath-nn_amd')]
import numpy as np, torch, arch_and_hypers as A

n = int(os.environ.get('BATCH', '128'))
kind = sys.argv[1] if len(sys.argv) > 1 else 'ac'
shares = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]
K = 8


def make(share, K=None):
    K = globals()['K'] if K is None else K
    nets, feeds = [], []
    g = torch.Generator().manual_seed(0)
    for i in range(K):
        mk = A.ac_chain(k_cpt=A.k_cpts[i % 8], seed=1234 + i) if kind == 'ac' else A.sr_chain(i + 1)
        net = mk((32, 32, 3), (10,))
        eng = net.engine()
        eng.co_share = share
        eng.x0[:n].copy_(torch.rand((n, 32, 32, 3), generator=g)); eng.y[:n].zero_(); eng.y[:n, i % 10] = 1
        nets.append(net)
        f = {net.x0: eng.x0[:n], net.y: eng.y[:n], net.mode: 'tr', net.λ_lrn: 0.1}
        if kind == 'ac':
            f[net.τ] = 1.0
        feeds.append(f)
    return nets, feeds


def wall(run, reps=60):
    for _ in range(6): run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


nets, feeds = make(1)
def serial():
    for net, f in zip(nets, feeds): net.train.run(f)
ms0 = wall(serial)
print('%s x %d, batch %d: serial loop %.1f us per round of all nets = %.0f img/s' % (kind, K, n, ms0 * 1e3, K * n / (ms0 * 1e-3)), flush=True)
del nets, feeds

for share in shares:
    try:
        nets, feeds = make(share)
        streams = [torch.cuda.Stream() for _ in range(K)]
        def side_by_side():
            for net, f, s in zip(nets, feeds, streams):
                with torch.cuda.stream(s):
                    net.train.run(f)
        ms = wall(side_by_side)
        print('  %d streams, grids budgeted for slots / %d: %.1f us per round = %.0f img/s, %.2fx serial'
              % (K, share, ms * 1e3, K * n / (ms * 1e-3), ms0 / ms), flush=True)
        del nets, feeds, streams
    except Exception as e:
        print('  share %d: %r' % (share, e), flush=True)

# groups of co-trained nets (lib/_co.py: CoGroups) side by side -- COGROUPS=G:share[:streams],...: TOTAL nets as G groups, each
# group's joint hipGraph on its own stream (round-robin over `streams`; groups of one net: any architecture)
if os.environ.get('COGROUPS'):
    from lib._co import CoGroups
    pre = [torch.cuda.Stream() for _ in range(int(os.environ.get('PRE', '0')))]     # (streams the process used before)
    for s_ in pre:
        with torch.cuda.stream(s_):
            torch.zeros(8, device='cuda').add_(1)
    torch.cuda.synchronize()
    TOTAL = int(os.environ.get('TOTAL', '8'))
    for spec in os.environ['COGROUPS'].split(','):
        v = [int(t) for t in spec.split(':')]
        G, share, S = v[0], v[1], (v[2] if len(v) > 2 else v[0])
        nets, feeds = make(1, TOTAL)
        if os.environ.get('ONEFIRST'):
            from lib._co import CoTrainer
            co = CoTrainer(nets)
            print('   one group first: %.1f us' % (wall(lambda: co.run(feeds)) * 1e3))
            del co
        base, extra = divmod(TOTAL, G)
        cg = CoGroups(nets, [base + (1 if i < extra else 0) for i in range(G)], streams=S, share=share)
        if os.environ.get('PIPE'):
            # through the input pipeline, as train-nets --co-train runs it: per group one record upload and one gather launch
            from lib.data import Dataset
            ds = Dataset.synthetic(n_tr=4096, n_ts=256, seed=1)
            ds.to_device('cuda:0')
            bound = [None] * TOTAL
            def bind(g, co, span): bound[span[0]:span[1]] = ds.bind_cotrainer(co, n)
            cg.on_group_streams(bind)
            fs = [{**f, net.x0: b[0], net.y: b[1]} for f, net, b in zip(feeds, nets, bound)]
            def step(g, co, span):
                ds.stage_cotrainer_draws(co)
                co.run(fs[span[0]:span[1]])
            ms = wall(lambda: cg.on_group_streams(step))
        else:
            ms = wall(lambda: cg.run(feeds))
        cg.join()
        if os.environ.get('PERGROUP') and cg.streams[0] is not None:
            torch.cuda.synchronize()
            e0 = [torch.cuda.Event(enable_timing=True) for _ in cg.streams]; e1 = [torch.cuda.Event(enable_timing=True) for _ in cg.streams]
            t0 = time.perf_counter()
            for e, s_ in zip(e0, cg.streams): e.record(s_)
            host = []
            for _ in range(20):
                h0 = time.perf_counter()
                (cg.on_group_streams(step) if os.environ.get('PIPE') else cg.run(feeds))
                host.append(time.perf_counter() - h0)
            for e, s_ in zip(e1, cg.streams): e.record(s_)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print('     per stream, 20 rounds: %s ms; host enqueue %.2f ms (%.0f us per round, max %.0f), drained after %.2f ms; stream ids %s'
                  % (' '.join('%.2f' % a.elapsed_time(b) for a, b in zip(e0, e1)), (t1 - t0) * 1e3, np.mean(host) * 1e6, max(host) * 1e6, (t2 - t0) * 1e3,
                     [s_.stream_id for s_ in cg.streams]))
            cg.join()
        print('  %d nets as %d groups on %d (%d) streams, grids for slots / %d: %.1f us per round = %.0f img/s, %.2fx serial'
              % (TOTAL, G, S, len(cg.streams), cg.share, ms * 1e3, TOTAL * n / (ms * 1e-3), ms0 / K * TOTAL / ms), flush=True)
        del cg, nets, feeds
