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
for spg in (4, 8):
    for _ in range(3): net.train.run_steps([feed] * spg)
    for steps in (20, 200):
        res = []
        for trial in range(12):
            for _ in range(2): net.train.run_steps([feed] * spg)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps // spg): net.train.run_steps([feed] * spg)
            for _ in range(steps % spg): net.train.run(feed)
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / steps * 1e3)
        print('spg %d, %3d-step region: ms/step min %.4f median %.4f max %.4f' % (spg, steps, min(res), float(np.median(res)), max(res)))
