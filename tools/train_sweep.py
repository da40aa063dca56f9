#!/usr/bin/env python3
"""The TRAINING step of the bench configuration (ac_chain(k_cpt=0), one net, one hipGraph per step) at 128 / 256 / 512 /
1 024 images per step, with the time of each launch family (one HIP-event pair around each run of consecutive launches of
one kind, whole steps launched eagerly) -- how much of the step at batch 128 is ramp rather than work.
    python tools/train_sweep.py [n ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'multipath-nn_amd')]
import numpy as np, torch, arch_and_hypers as A

F_TRAIN, PEAK = 123.024e6, 157.3e12
sizes = [int(a) for a in sys.argv[1:]] or [128, 256, 512, 1024]
print('%6s %10s %11s %8s | %s' % ('batch', 'us/step', 'img/s', 'of peak', 'per family: launches, us, TFLOP/s'))
base = None
for n in sizes:
    net = A.ac_chain(k_cpt=0.0, seed=1234)((32, 32, 3), (10,))
    eng = net.engine()
    eng._ensure_capacity(n)
    g = torch.Generator().manual_seed(0)
    eng.x0[:n].copy_(torch.rand((n, 32, 32, 3), generator=g)); eng.y[:n].zero_(); eng.y[:n, 0] = 1
    feed = {net.x0: eng.x0[:n], net.y: eng.y[:n], net.mode: 'tr', net.λ_lrn: 0.1, net.τ: 1.0}
    for _ in range(8): net.train.run(feed)
    torch.cuda.synchronize()
    st = torch.cuda.current_stream()
    K = 20
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    evs[0].record(st)
    for k in range(K):
        for _ in range(10): net.train.run(feed)
        evs[k + 1].record(st)
    torch.cuda.synchronize()
    us = float(np.median([evs[k].elapsed_time(evs[k + 1]) / 10 for k in range(K)])) * 1e3
    fam = eng.time_family_blocks('tr', n, reps=10)
    eng.mark_dirty()
    parts = []
    for what in ('fwd_group', 'lin_fwd', 'exit_tail_fwd', 'route', 'exit_tail_bwd', 'lin_bwd', 'bwd_scale', 'backward_finish'):
        if what in fam:
            cnt, fl, ms = fam[what]
            parts.append('%s %d: %.1f us%s' % (what, cnt, ms * 1e3, (' %.1f TF' % (fl / (ms * 1e-3) / 1e12)) if fl else ''))
    if base is None:
        base = us / n
    print('%6d %10.1f %11.0f %8.3f | %s' % (n, us, n / (us * 1e-6), n / (us * 1e-6) * F_TRAIN / PEAK, '; '.join(parts)), flush=True)
    print('%6s   %.2f us per image (%.2fx the rate per image at batch %d)' % ('', us / n, base / (us / n), sizes[0]), flush=True)
    del net, eng
