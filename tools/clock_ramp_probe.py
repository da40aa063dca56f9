"""Does a SHORT timed region right after process start read low?  Consecutive 20-step regions (five replays of the four-step
graph, a synchronisation before and after each) from the first replay after capture on; then the same after 1 s of idle."""
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
for _ in range(3): net.train.run_steps([feed] * 4)


def regions(k, label):
    out = []
    for _ in range(k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): net.train.run_steps([feed] * 4)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 20 * 1e3)
    if label is not None:
        print(label, ' '.join('%.4f' % v for v in out), flush=True)
    return out


regions(16, 'right after capture:')
for idle in (0.002, 0.005, 0.02, 0.05, 0.1, 0.3, 1.0, 5.0):
    firsts = []
    for _ in range(5):
        regions(2, None)
        time.sleep(idle)
        firsts.append(regions(1, None)[0])
    print('first 20-step region after %5.0f ms of idle: %s' % (idle * 1e3, ' '.join('%.4f' % v for v in firsts)), flush=True)
