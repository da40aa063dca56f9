"""In-situ per-launch table of one training step (whole steps run eagerly, HIP events around every launch)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd'))
import torch, arch_and_hypers as A
net = A.ac_chain(k_cpt=0.0, seed=1234)((32, 32, 3), (10,))
eng = net.engine()
n = 128
eng.x0[:n].uniform_(); eng.y[:n].zero_(); eng.y[:n, 0] = 1
feed = {net.x0: eng.x0[:n], net.y: eng.y[:n], net.mode: 'tr', net.λ_lrn: 0.1, net.τ: 1.0}
for _ in range(3): net.train.run(feed)
ops = eng.time_step_ops('tr', n, reps=20)
tot = sum(o[3] for o in ops)
print('sum %.1f us over %d launches (per-launch event pairs add ~1.5 us of host latency each)' % (tot * 1e3, len(ops)))
for what, tag, fl, ms in ops:
    print('%-16s %-34s %8.1f us  %6.2f TFLOP/s' % (what, tag, ms * 1e3, fl / (ms * 1e-3) / 1e12 if fl else 0))
