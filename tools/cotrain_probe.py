#!/usr/bin/env python3
"""Aggregate training rate of K co-trained `ac_chain(k_cpt=k)` nets (lib/_co.py), batch 128 each, against the serial rate
(the same nets stepped one after another, one hipGraph each): python tools/cotrain_probe.py [K ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'multipath-nn_amd')]
import numpy as np, torch, arch_and_hypers as A
from lib._co import CoTrainer

n = int(os.environ.get('BATCH', '128'))
Ks = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]


def timed(run, reps=200, chunk=10):
    for _ in range(8): run()
    torch.cuda.synchronize()
    st = torch.cuda.current_stream()
    k = reps // chunk
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
    evs[0].record(st)
    for i in range(k):
        for _ in range(chunk): run()
        evs[i + 1].record(st)
    torch.cuda.synchronize()
    return float(np.median([evs[i].elapsed_time(evs[i + 1]) / chunk for i in range(k)]))


def make(K):
    nets, feeds = [], []
    g = torch.Generator().manual_seed(0)
    for i in range(K):
        net = A.ac_chain(k_cpt=A.k_cpts[i % 8], seed=1234 + i)((32, 32, 3), (10,))
        eng = net.engine()
        eng.x0[:n].copy_(torch.rand((n, 32, 32, 3), generator=g)); eng.y[:n].zero_(); eng.y[:n, i % 10] = 1
        nets.append(net)
        feeds.append({net.x0: eng.x0[:n], net.y: eng.y[:n], net.mode: 'tr', net.λ_lrn: 0.1, net.τ: 1.0})
    return nets, feeds

serial = None
for K in Ks:
    nets, feeds = make(K)
    if serial is None:
        serial = timed(lambda: nets[0].train.run(feeds[0]))
        print('solo step: %.1f us = %.0f img/s' % (serial * 1e3, n / (serial * 1e-3)), flush=True)
    co = CoTrainer(nets)
    ms = timed(lambda: co.run(feeds), reps=100 if K > 2 else 200)
    print('K = %d co-trained: %.1f us per step of all nets = %.0f img/s aggregate, %.2fx the serial rate (%.1f us per net-step)'
          % (K, ms * 1e3, K * n / (ms * 1e-3), K * serial / ms, ms * 1e3 / K), flush=True)
    if os.environ.get('TABLE'):
        prog = co._program(n)
        st = torch.cuda.current_stream()
        for e in co.engs: e.mark_dirty()
        co.use_graph = False
        tot = [0.0] * len(prog['ops'])
        for rep in range(6):
            for e in co.engs: e._begin(True)
            evs = []
            for op in prog['ops']:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st); op(st.cuda_stream); e1.record(st); evs.append((e0, e1))
            torch.cuda.synchronize()
            if rep:
                for j, (e0, e1) in enumerate(evs): tot[j] += e0.elapsed_time(e1) / 5
        for op, t in zip(prog['ops'], tot):
            print('   %-16s %-44s %7.1f us  %6.1f TFLOP/s' % (op.what, op.tag[:44], t * 1e3, op.flops / (t * 1e-3) / 1e12 if t else 0))
        for e in co.engs: e.mark_dirty()
    del co, nets, feeds

if os.environ.get('PIPE'):
    # the same with the input pipeline: every net draws its own batch per step (the reference's draws, one numpy stream),
    # K record uploads and K on-device gathers at the head of the joint graph
    from lib.data import Dataset
    for K in Ks:
        nets, feeds = make(K)
        ds = Dataset.synthetic(n_tr=4096, n_ts=256, seed=1)
        engs = [net.engine() for net in nets]
        co = CoTrainer(nets)
        bound = ds.bind_cotrainer(co, n) if os.environ.get('PIPE') != 'each' else [ds.bind_engine(e, n) for e in engs]
        fs = [{net.x0: x0, net.y: y, net.mode: 'tr', net.λ_lrn: 0.1, net.τ: 1.0} for net, (x0, y) in zip(nets, bound)]

        def step():
            if os.environ.get('PIPE') != 'each':
                ds.stage_cotrainer_draws(co)
            else:
                for e in engs:
                    ds.stage_training_draws(n, eng=e)
            co.run(fs)
        ms = timed(step, reps=100)
        print('K = %d co-trained WITH the input pipeline: %.1f us per joint step = %.0f img/s' % (K, ms * 1e3, K * n / (ms * 1e-3)), flush=True)
        del co, nets, feeds
