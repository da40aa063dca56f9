"""Dense vs routed evaluation throughput over batch sizes (exit fractions 1/8 each)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd'))
import torch, numpy as np, arch_and_hypers as A
import bench
net = A.ac_chain(k_cpt=0.0, seed=1234)((32, 32, 3), (10,))
eng = net.engine()
rng = np.random.default_rng(5)
for l in net.layers:                      # the last router map starts at zero: give the routers something to decide on
    if l.router is not None:
        w = l.router.comps[-1].params.w
        w.assign(rng.standard_normal(w.shape) * 0.5)
for nb in [int(a) for a in sys.argv[1:]] or [1024, 4096, 8192]:
    x, y = bench.synthetic(nb, 1, 'cuda:0')
    eng._ensure_capacity(nb, train=False)
    eng.x0[:nb].copy_(x); eng.y[:nb].copy_(y)
    feed = {net.x0: eng.x0[:nb], net.y: eng.y[:nb]}
    bench.set_exit_fractions(net, feed, nb, [1 / 8] * 7)
    res = {}
    if os.environ.get('PREFIX_SWEEP'):
        # routed evaluation with the convs of the blocks above depth d0 run on every sample (lib/_plan.py:_program_ev)
        line = []
        for d0 in (False, 1, 2, 3, 4, 5, 6, 7, 8):
            for _ in range(3): net.eval(feed, routed=d0)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(20): net.eval(feed, routed=d0)
            torch.cuda.synchronize(); line.append((d0, (time.perf_counter() - t) / 20 * 1e3))
        print('batch %6d: ' % nb + '  '.join('%s %.3f' % ('dense' if d0 is False else 'd0=%d' % d0, t) for d0, t in line) + '  (ms)', flush=True)
        continue
    for routed in (False, True, 'auto'):
        for _ in range(3): net.eval(feed, routed=routed)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): net.eval(feed, routed=routed)
        torch.cuda.synchronize(); res[routed] = (time.perf_counter() - t) / 10 * 1e3
    print('batch %6d: dense %.3f ms (%.2f M img/s)  routed %.3f ms (%.2f M img/s)  x%.2f   routed=auto %.3f ms (x%.2f)' % (
        nb, res[False], nb / res[False] / 1e3, res[True], nb / res[True] / 1e3, res[False] / res[True], res['auto'], res[False] / res['auto']))
    if nb == 4096:
        print('   exit histogram', [round(float(nd.layer.p_ev.mean()), 3) for nd in eng.leaves])
        prog = eng.program('ev', nb, routed=True)
        st = torch.cuda.current_stream()
        tot = [0.0] * len(prog['fwd'])
        for rep in range(6):
            eng._begin(False)                      # (clears the sample counts: the lists are appended to)
            evs = []
            for op in prog['fwd']:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st); op(st.cuda_stream); e1.record(st)
                evs.append((e0, e1))
            torch.cuda.synchronize()
            if rep:
                for k, (e0, e1) in enumerate(evs): tot[k] += e0.elapsed_time(e1) / 5
        for op, t in zip(prog['fwd'], tot):
            print('   routed %-12s %-30s %8.1f us' % (op.what, op.tag[:30], t * 1e3))
