"""What does the data-parallel step STRUCTURE cost on one GPU (1-rank RCCL group)?
  single   : one hipGraph per step (no DP)
  sections : section graphs + optimizer graph, collectives replaced by a no-op
  rccl3    : section graphs + async RCCL all-reduce per bucket (the shipped DP step)
  rccl1    : one bucket (MPNN_DP_BUCKETS=1): fwd+bwd graph, ONE all-reduce of G, optimizer graph
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd'))
import numpy as np, torch, torch.distributed as dist, arch_and_hypers as A
from lib import _dp
import bench

def build():
    net = A.ac_chain(k_cpt=0.0, seed=1234)((32, 32, 3), (10,))
    eng = net.engine()
    n = 128
    x0, y = bench.synthetic(n, 0, 'cuda:0')
    eng.x0[:n].copy_(x0); eng.y[:n].copy_(y)
    feed = {net.x0: eng.x0[:n], net.y: eng.y[:n], net.mode: 'tr', net.λ_lrn: 0.1, net.τ: 1.0}
    return net, eng, feed

def timeit(net, feed, tag):
    for _ in range(6): net.train.run(feed)
    torch.cuda.synchronize()
    ms = bench.time_replays(lambda: net.train.run(feed), 300)
    print('%-10s %.1f us/step' % (tag, ms * 1e3), flush=True)
    return ms

net, eng, feed = build()
base = timeit(net, feed, 'single')
_dp.init(backend='nccl', force=True)
mode = sys.argv[1] if len(sys.argv) > 1 else 'all'
if mode in ('all', 'whole'):
    net, eng, feed = build()
    _dp.attach(net, force=True)
    timeit(net, feed, 'whole3')
    os.environ['MPNN_DP_BUCKETS'] = '1'
    net, eng, feed = build()
    _dp.attach(net, force=True)
    timeit(net, feed, 'whole1')
    os.environ['MPNN_DP_BUCKETS'] = '3'
os.environ['MPNN_DP_ONE_GRAPH'] = '0'
if mode in ('all', 'sections'):
    net, eng, feed = build()
    _dp.attach(net, force=True)
    eng.allreduce = lambda flat: None
    timeit(net, feed, 'sections')
if mode in ('all', 'rccl3'):
    net, eng, feed = build()
    _dp.attach(net, force=True)
    timeit(net, feed, 'rccl3')
if mode in ('all', 'rccl1'):
    os.environ['MPNN_DP_BUCKETS'] = '1'
    net, eng, feed = build()
    _dp.attach(net, force=True)
    timeit(net, feed, 'rccl1')
dist.destroy_process_group()
