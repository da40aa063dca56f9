import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'multipath-nn_amd')]
import numpy as np, torch, arch_and_hypers as A
import bench
n = 128
net = A.ac_chain(k_cpt=0.0, seed=1234)((32, 32, 3), (10,))
eng = net.engine()
x0, y = bench.synthetic(n, 0, 'cuda:0')
eng.x0[:n].copy_(x0); eng.y[:n].copy_(y)
feed = {net.x0: eng.x0[:n], net.y: eng.y[:n], net.mode: 'tr', net.λ_lrn: A.λ_lrn(0), net.τ: A.τ_ds(0)}
for _ in range(3): net.train.run(feed)
TRIALS = int(os.environ.get('TRIALS', '12'))
for spg in (4, 8):
    for _ in range(3): net.train.run_steps([feed] * spg)
    for steps in (20, 200):
        rem = steps % spg
        if rem > 1:
            for _ in range(3): net.train.run_steps([feed] * rem)       # (the remainder as a K-step graph of its own)
        res, host = [], []
        for trial in range(TRIALS):
            for _ in range(2): net.train.run_steps([feed] * spg)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps // spg): net.train.run_steps([feed] * spg)
            if rem > 1: net.train.run_steps([feed] * rem)
            elif rem: net.train.run(feed)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / steps * 1e3)
            host.append((t1 - t0) / steps * 1e3)
        print('spg %d, %3d-step region: ms/step min %.4f median %.4f p90 %.4f max %.4f   (host returns after %.4f ms/step, median)'
              % (spg, steps, min(res), float(np.median(res)), float(np.percentile(res, 90)), max(res), float(np.median(host))))
