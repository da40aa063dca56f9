#!/usr/bin/env python3
"""Host time per call of the single-net training entry points (no synchronisation inside the loop: what the host needs to
enqueue a step) beside the GPU time per step -- is the ~8 us between two graph replays the host's or the runtime's?
    python tools/host_step_probe.py"""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'multipath-nn_amd')]
import numpy as np, torch, arch_and_hypers as A

n = 128
net = A.ac_chain(k_cpt=0.0, seed=1234)((32, 32, 3), (10,))
eng = net.engine()
eng._ensure_capacity(n)
eng.x0[:n].copy_(torch.rand((n, 32, 32, 3))); eng.y[:n].zero_(); eng.y[:n, 3] = 1
feed = {net.x0: eng.x0[:n], net.y: eng.y[:n], net.mode: 'tr', net.λ_lrn: 0.1, net.τ: 1.0}

def measure(call, steps_per_call, label, reps=300):
    for _ in range(10): call()
    torch.cuda.synchronize()
    h = []
    t0 = time.perf_counter()
    for _ in range(reps):
        a = time.perf_counter(); call(); h.append(time.perf_counter() - a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    h = np.array(h) * 1e6
    print('%-34s host per call: median %.0f us (p95 %.0f); enqueue loop %.1f us per step, drained %.1f us per step'
          % (label, np.median(h), np.percentile(h, 95), (t1 - t0) / reps / steps_per_call * 1e6, (t2 - t0) / reps / steps_per_call * 1e6), flush=True)

measure(lambda: net.train.run(feed), 1, 'net.train.run (one step per graph)')
measure(lambda: net.train.run_steps([feed] * 4), 4, 'run_steps, 4 steps per graph')
measure(lambda: net.train.run_steps([feed] * 8), 8, 'run_steps, 8 steps per graph')
if os.environ.get('PROFILE'):
    pr = cProfile.Profile(); pr.enable()
    for _ in range(300): net.train.run(feed)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
