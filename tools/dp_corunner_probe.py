"""Does the data-parallel step survive a kernel that runs BESIDE it?  (one GPU, no process group)

Under data parallelism RCCL's all-reduce kernels run on the process group's stream while the backward pass goes on
(lib/_plan.py: a bucket's collective is issued where the bucket becomes final).  The backward launches use persistent
grids fitted to the workgroups that are resident at once; a co-running kernel takes some of those slots, and a grid
whose last workgroups only start when others exit can take twice as long (tools/overlap_probe.py: 37 -> 263 us).
No multi-GPU box is available to the build, so the collective is replaced by a stand-in: `k` workgroups x 512 threads
that hold their slots for `T` microseconds (mpnn_debug_spin), launched on a side stream at every bucket point of the
ONE-graph data-parallel step -- exactly where and how the RCCL kernels enter the graph.  Measured per (k, T):

  res0   : grids fitted to every compute unit (MPNN_DP_RESERVE_CUS=0)
  resK   : the trunk backward leaves `reserve` compute units free (mpnn_set_reserved_cus; default of the engine)
  hidden : co-runners at the buckets that overlap the backward pass only (exit, mid); the stand-in of the last bucket
           (`end`, which nothing can hide) has T = 0 -- this column isolates CONTENTION
  all    : the stand-in runs T us at every bucket, the last one fully exposed (= contention + T)

    python tools/dp_corunner_probe.py [--reserve 16] [--reps 300]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd'))
import torch
import arch_and_hypers as A
import bench


class _Handle:
    def __init__(self, side):
        self.side = side

    def wait(self):
        torch.cuda.current_stream().wait_stream(self.side)


def install_corunner(eng, k, T, T_last=None):
    """Replace the collective of a (1-rank) data-parallel engine by k spinning workgroups of 512 threads for T us."""
    side = torch.cuda.Stream(device=eng.dev)
    last = list(eng.dp_buckets.values())[-1]

    def corunner(flat):
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        is_last = flat.data_ptr() == eng.G[last[0]:].data_ptr()
        us = (T if T_last is None else T_last) if is_last else T
        if k > 0:
            eng.lib.mpnn_debug_spin(k, 512, float(us), side.cuda_stream)
        return _Handle(side)
    eng.world, eng.allreduce, eng.allreduce_capturable, eng.dp_agree = 1, corunner, True, None
    eng._graphs.clear()
    eng._keep.append(side)
    return eng


BUCKETS = 3          # the probe is about the BUCKETED form unless told otherwise (--buckets 1: the shipped default)


def build(reserve, n=128, bucket_opt=None, buckets=None):
    os.environ['MPNN_DP_BUCKETS'] = str(BUCKETS if buckets is None else buckets)
    net = A.ac_chain(k_cpt=0.0, seed=1234)((32, 32, 3), (10,))
    eng = net.engine()
    del os.environ['MPNN_DP_BUCKETS']
    eng.dp_reserve_cus = reserve
    if bucket_opt is not None:
        eng.dp_bucket_opt = bool(bucket_opt)
    x0, y = bench.synthetic(n, 0, 'cuda:0')
    eng.x0[:n].copy_(x0); eng.y[:n].copy_(y)
    feed = {net.x0: eng.x0[:n], net.y: eng.y[:n], net.mode: 'tr', net.λ_lrn: 0.1, net.τ: 1.0}
    return net, eng, feed


def timeit(net, feed, reps):
    for _ in range(6):
        net.train.run(feed)
    torch.cuda.synchronize()
    return bench.time_replays(lambda: net.train.run(feed), reps) * 1e3


def measure(reserve, k, T, T_last, reps, n=128, bucket_opt=None, buckets=None):
    net, eng, feed = build(reserve, n, bucket_opt, buckets)
    install_corunner(eng, k, T, T_last)
    if os.environ.get('PROBE_EAGER'):              # eager launches on real streams instead of the step graph
        eng.use_graph = False
    us = timeit(net, feed, reps)
    if eng.use_graph:
        key = [q for q in eng._graphs if q[0] == 'tr' and q[2]][0]
        assert eng._graphs[key][1] == 'whole', 'the probe measures the one-graph step'
    del net, eng
    return us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reserve', type=int, nargs='*', default=[16])
    ap.add_argument('--reps', type=int, default=300)
    ap.add_argument('--ks', type=int, nargs='*', default=[0, 4, 8, 16, 32])
    ap.add_argument('--ts', type=float, nargs='*', default=[20.0, 40.0])
    ap.add_argument('--structure', action='store_true',
                    help='what the FORM of the step costs with no co-runner at all (k = 0): per-bucket updates on a side '
                         'stream on / off, reservation on / off')
    ap.add_argument('--buckets', type=int, default=3)
    ap.add_argument('--one', type=float, nargs=4, metavar=('BUCKET_OPT', 'RESERVE', 'K', 'T'),
                    help='one configuration (last bucket T = 0), one line: for environment sweeps')
    args = ap.parse_args()
    global BUCKETS
    BUCKETS = args.buckets
    if args.one:
        bo, r, k, T = args.one
        v = [measure(int(r), int(k), T, 0.0, args.reps, bucket_opt=int(bo)) for _ in range(2)]
        print('%s bucket_opt %d reserve %d k %d T %.0f: %s us' % (os.environ.get('TAG', ''), bo, r, k, T, ' / '.join('%.1f' % x for x in v)))
        return
    if args.structure:
        net, eng, feed = build(0)
        single = timeit(net, feed, args.reps)
        print('single-process step: %.1f us' % single)
        del net, eng
        for bo in (0, 1):
            for r in [0] + args.reserve:
                for k, T in ((0, 0.0), (16, 40.0)):
                    v = [measure(r, k, T, 0.0, args.reps, bucket_opt=bo) for _ in range(2)]
                    print('per-bucket update %d  reserve %2d  co-runner k=%2d T=%2.0f (last bucket T=0): %s us  (%.3f)'
                          % (bo, r, k, T, ' / '.join('%.1f' % x for x in v), min(v) / single), flush=True)
        return
    net, eng, feed = build(0)
    single = timeit(net, feed, args.reps)
    print('single-process step (one hipGraph, no data parallelism): %.1f us' % single)
    del net, eng
    print('\nONE bucket (the shipped default): the stand-in sits between the launch that ends the backward pass and the optimizer')
    for k, T in ((0, 0.0), (16, 0.0), (16, 20.0), (16, 40.0), (32, 40.0)):
        v = measure(0, k, T, None, args.reps, buckets=1)
        print('  k %2d  T %2.0f us : %7.1f us (%.3f of the single-process step; minus T: %.1f us)' % (k, T, v, v / single, v - T), flush=True)
    net, eng, feed = build(0)
    print('\nTHREE buckets: %s' % {k: (hi - lo) * 4 for k, (lo, hi) in eng.dp_buckets.items()}, '(bytes)')
    del net, eng
    cols = ['res0'] + ['res%d' % r for r in args.reserve]
    print('%4s %5s | %s | %s' % ('k', 'T us', ' '.join('%14s' % ('hidden ' + c) for c in cols), ' '.join('%14s' % ('all ' + c) for c in cols)))
    for T in args.ts:
        for k in args.ks:
            hid = [measure(r, k, T, 0.0, args.reps) for r in [0] + args.reserve]
            al = [measure(r, k, T, None, args.reps) for r in [0] + args.reserve]
            fmt = lambda v: '%7.1f (%.3f)' % (v, v / single)
            print('%4d %5.0f | %s | %s' % (k, T, ' '.join(fmt(v) for v in hid), ' '.join(fmt(v) for v in al)), flush=True)


if __name__ == '__main__':
    main()
