import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'multipath-nn_amd'))
import torch, arch_and_hypers as A
net = A.ac_chain(k_cpt=0.0, seed=1234)((32, 32, 3), (10,))
eng = net.engine(); n = 128
eng.x0[:n].uniform_(); eng.y[:n].zero_(); eng.y[:n, 0] = 1
feed = {net.x0: eng.x0[:n], net.y: eng.y[:n], net.mode: 'tr', net.λ_lrn: 0.0, net.τ: 1.0}
net.train.run(feed)
eng.multi_stream = True          # separate launches per member
orig = eng._wsplit
eng._wsplit = lambda b, i, n, fused=False: orig(b, i, n, fused=True)   # same splits/groups as the fused path
ops = eng.time_ops('tr', n, reps=20)
for what, tag, fl, ms in ops:
    if what in ('wgrad', 'dgrad_horz', 'dgrad_vert', 'bn_bwd_apply') and ('h4' in tag or 'h8 64' in tag or tag == ''):
        print('%-12s %-18s %7.1f us %6.1f TF' % (what, tag, ms * 1e3, fl / (ms * 1e-3) / 1e12 if fl else 0))
