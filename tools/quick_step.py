"""Steady-state step time of the bench configuration (one hipGraph per step), HIP events around chunks of ten:
the quick A/B number while iterating on kernels or environment settings.

    python tools/quick_step.py [steps]      ->  'step_us <median> <mean> <img/s>'
"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd'))
import numpy as np, torch, arch_and_hypers as A
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
net = A.ac_chain(k_cpt=0.0, seed=1234)((32, 32, 3), (10,))
eng = net.engine()
n = 128
g = torch.Generator().manual_seed(0)
eng.x0[:n].copy_(torch.rand((n, 32, 32, 3), generator=g)); eng.y[:n].zero_(); eng.y[:n, 0] = 1
feed = {net.x0: eng.x0[:n], net.y: eng.y[:n], net.mode: 'tr', net.λ_lrn: 0.1, net.τ: 1.0}
SPG = int(os.environ.get('SPG', '1'))          # training steps per hipGraph replay (Engine.run_steps); 1: one graph per step
def ten():
    if SPG > 1:
        for _ in range(10 // SPG): net.train.run_steps([feed] * SPG)
        for _ in range(10 % SPG): net.train.run(feed)
    else:
        for _ in range(10): net.train.run(feed)
for _ in range(8): net.train.run(feed)
for _ in range(3): ten()
torch.cuda.synchronize()
st = torch.cuda.current_stream()
K = steps // 10
evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
evs[0].record(st)
for k in range(K):
    ten()
    evs[k + 1].record(st)
torch.cuda.synchronize()
per = np.array([evs[k].elapsed_time(evs[k + 1]) / 10 for k in range(K)]) * 1e3
print('step_us %.1f %.1f %.0f' % (np.median(per), per.mean(), n / (np.median(per) * 1e-6)), os.environ.get('TAG', ''))
