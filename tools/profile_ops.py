"""Per-launch timing table of one training step (HIP events on the launch stream)."""
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
ops = eng.time_ops('tr', n, reps=30)
tot = sum(o[3] for o in ops)
agg = {}
for what, tag, fl, ms in ops:
    a = agg.setdefault(what, [0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += fl
print('total %.1f us over %d launches' % (tot * 1e3, len(ops)))
for k, (c, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-16s x%-3d %8.1f us %5.1f%%  %6.2f TFLOP/s' % (k, c, ms * 1e3, 100 * ms / tot, fl / (ms * 1e-3) / 1e12 if ms else 0))
print()
for what, tag, fl, ms in ops:
    print('%-16s %-18s %8.1f us  %6.2f TFLOP/s' % (what, tag, ms * 1e3, fl / (ms * 1e-3) / 1e12 if fl else 0))

print('\n-- ev mode (moving-average BN, no statistics atomics) --')
net.eval({net.x0: eng.x0[:n], net.y: eng.y[:n]})
for what, tag, fl, ms in eng.time_ops('ev', n, reps=30):
    print('%-16s %-18s %8.1f us  %6.2f TFLOP/s' % (what, tag, ms * 1e3, fl / (ms * 1e-3) / 1e12 if fl else 0))
