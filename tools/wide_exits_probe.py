#!/usr/bin/env python3
"""Exit path beyond the tuned kernels' limits (csrc/exit_gen.hip): the training step of a 4-block actor chain at batch
128 with the shipped exits (10 classes, 16-16 routers: tuned kernels) against 100 classes and 32-32 routers (any-width
kernels), step time and per-launch table.   python tools/wide_exits_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'multipath-nn_amd'), os.path.join(ROOT, 'tests')]
import numpy as np, torch, arch_and_hypers as A
from lib.net_types import ActorNet
from test_net_parity import _wide_chain

n = 128


def build(widths, n_cls, n_blocks):
    net = _wide_chain(ActorNet, widths, n_blocks=n_blocks, k_cpt=1.6e-8)((32, 32, 3), (n_cls,))
    eng = net.engine()
    eng.init_params(5)
    g = torch.Generator().manual_seed(0)
    eng.x0[:n].copy_(torch.rand((n, 32, 32, 3), generator=g)); eng.y[:n].zero_(); eng.y[:n, 0] = 1
    return net, eng, {net.x0: eng.x0[:n], net.y: eng.y[:n], net.mode: 'tr', net.λ_lrn: 0.05, net.τ: 1.0}


def step_us(net, feed, reps=200):
    for _ in range(8): net.train.run(feed)
    torch.cuda.synchronize()
    st = torch.cuda.current_stream()
    K = reps // 10
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    evs[0].record(st)
    for k in range(K):
        for _ in range(10): net.train.run(feed)
        evs[k + 1].record(st)
    torch.cuda.synchronize()
    return float(np.median([evs[k].elapsed_time(evs[k + 1]) / 10 for k in range(K)])) * 1e3

for nb in (4, 8):
    base = None
    for name, widths, n_cls in (('10 classes, 16-16 routers', (16, 16), 10), ('100 classes, 32-32 routers', (32, 32), 100),
                                ('100 classes, 16-16 routers', (16, 16), 100), ('10 classes, 48-16 routers', (48, 16), 10)):
        net, eng, feed = build(widths, n_cls, nb)
        us = step_us(net, feed)
        base = base or us
        print('%d blocks, %-28s generic=%-5s step %7.1f us  (%.2fx)' % (nb, name, eng.generic_exits, us, us / base), flush=True)
        if os.environ.get('TABLE') and nb == 4:
            for what, tag, fl, ms in eng.time_step_ops('tr', n, reps=10):
                if what not in ('fwd_group', 'bwd_scale'):
                    print('      %-16s %8.1f us' % (what, ms * 1e3))
            eng.mark_dirty()
        del net, eng
