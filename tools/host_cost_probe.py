#!/usr/bin/env python3
"""Host time per round of a co-trained experiment through the input pipeline (cProfile): the GPU round is ~1.9 ms for 8
nets, the host must stay well below it.     python tools/host_cost_probe.py [nets]"""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'multipath-nn_amd')]
import numpy as np, torch, arch_and_hypers as A
from lib._co import CoGroups
from lib.data import Dataset

K, n = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 128
nets = [A.ac_chain(k_cpt=A.k_cpts[i % 8], seed=1234 + i)((32, 32, 3), (10,)) for i in range(K)]
ds = Dataset.synthetic(n_tr=4096, n_ts=256, seed=1)
ds.to_device('cuda:0')
cg = CoGroups.plan(nets, streams=4)
bound = [None] * K
def bind(g, co, span): bound[span[0]:span[1]] = ds.bind_cotrainer(co, n)
cg.on_group_streams(bind)
t = [0]
def round_():
    feeds = [{net.x0: b[0], net.y: b[1], net.mode: 'tr', net.λ_lrn: A.λ_lrn(t[0]), net.τ: A.τ_ds(t[0])} for net, b in zip(nets, bound)]
    t[0] += 1
    def step(g, co, span):
        ds.stage_cotrainer_draws(co)
        co.run(feeds[span[0]:span[1]])
    cg.on_group_streams(step)
for _ in range(10): round_()
torch.cuda.synchronize()
h = []
for _ in range(200):
    t0 = time.perf_counter(); round_(); h.append(time.perf_counter() - t0)
torch.cuda.synchronize()
h = np.array(h) * 1e6
print('host enqueue per round: median %.0f us, mean %.0f, p95 %.0f, max %.0f' % (np.median(h), h.mean(), np.percentile(h, 95), h.max()))
pr = cProfile.Profile()
pr.enable()
for _ in range(100): round_()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
